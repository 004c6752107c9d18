// dcp_kernels.hip -- gfx950 kernels of the profile-HMM scan engine.
//
//  expand_tables_kernel   frame-state emission tables (imm frame state,
//                         SURVEY Appendix A) for every node of every profile,
//                         written code-major / node-contiguous.
//  viterbi_rowsweep_kernel<R>  exact null + alt Viterbi of one (profile, query)
//                         pair per wavefront: the recursion of SURVEY
//                         Appendix B = what imm_dp_viterbi computes for
//                         thread_run (src/server/scan_thread.c:99-123).
//
// Arithmetic contract (bit-exact with the CPU oracle's float32 build): every
// candidate is formed as (predecessor + transition) + emission in IEEE float32,
// combined with max only; no multiplies, no FMA, no reassociation.
#include "dcp_kernels.h"

#include <hip/hip_runtime.h>
#include <algorithm>

namespace
{

__device__ __forceinline__ float neg_inf() { return -__builtin_inff(); }

// Read-only inputs at wave-uniform addresses (sequence words, background / insert emissions, the
// query's special transitions) are read through the constant address space: they become scalar loads
// into SGPRs -- free operands of the VALU -- instead of 64 lanes loading the same dword.
typedef float const __attribute__((address_space(4))) cfloat;
typedef uint32_t const __attribute__((address_space(4))) cu32;
__device__ __forceinline__ cfloat *as_const(float const *p) { return (cfloat *)(unsigned long long)p; }
__device__ __forceinline__ cu32 *as_const(uint32_t const *p) { return (cu32 *)(unsigned long long)p; }

// ---- cross-lane helpers (wave64 DPP) ---------------------------------------
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float old_value, float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old_value),
                                           __builtin_bit_cast(int, v), CTRL,
                                           ROW_MASK, BANK_MASK, false));
}

// value of lane-1 (lane 0 receives `first`)
__device__ __forceinline__ float lane_shr1(float v, float first)
{
    return dpp_mov<0x138 /*wave_shr:1*/, 0xf, 0xf>(first, v);
}

// Cross-lane maxima and shifted adds are written as ONE VALU instruction with a DPP operand
// (v_max_f32_dpp / v_add_f32_dpp) where the compiler would emit three (constant for the lanes
// without a source, v_mov_b32_dpp, the operation).  A lane without a source keeps the destination's
// old value; `s_nop 1` covers the two wait states a DPP read of a just-written VGPR needs (the
// compiler's hazard recogniser does not look inside asm statements).

// max over each group of four lanes, returned in all four
__device__ __forceinline__ float quad_max(float v)
{
    float r;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(r)
        : "v"(v));
    return r;
}

// (x of lane - 1) + tr; lane 0 of the wavefront: `first` + tr when HAS_FIRST, else 0 + tr -- used where
// lane 0 is the profile's first node, whose incoming transitions are -inf (dcp_gpu_db_upload).
template <bool HAS_FIRST> __device__ __forceinline__ float shr1_add(float x, float first, float tr)
{
    float r;
    if constexpr (HAS_FIRST)
    {
        r = first + tr;
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf"
            : "+v"(r)
            : "v"(x), "v"(tr));
    }
    else
    {
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
            : "=v"(r)
            : "v"(x), "v"(tr));
    }
    return r;
}

// max over the 64 lanes, returned wave-uniform
__device__ __forceinline__ float wave_max(float v)
{
    float r;
    // first step with bound_ctrl:0: a lane without a source (lanes 0 of a row here, 0..2 after the next two
    // steps) takes 0 for it and holds garbage from then on, but only lanes 3, 7, 11, 15 of a row feed the
    // steps that follow, and all their sources exist -- so the destination needs no initial copy of v
    asm("s_nop 1\n\t" // v may have been written by the instruction before (DPP source: two wait states)
        "v_max_f32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_max_f32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "=&v"(r)
        : "v"(v));
    return __builtin_bit_cast(
        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 63));
}

// max over each PART of the wavefront -- 64 / K consecutive lanes, K = 2 or 4 -- returned in every lane of the part
// (viterbi_mp_kernel: K profiles per wavefront).  Four rotations inside the rows of 16 lanes make every lane hold its
// row's maximum (cheaper than wave_max: no lane read); for K = 2 row 1 / 3 then takes row 0 / 2's over (row_bcast:15),
// lanes 31 and 63 are read, and each half picks its own.
template <int K> __device__ __forceinline__ float part_max(float v)
{
    float r;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "=&v"(r)
        : "v"(v));
    if constexpr (K == 4) return r;
    else
    {
        static_assert(K == 2, "two or four profiles per wavefront");
        asm("s_nop 1\n\t"
            "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"
            : "+v"(r));
        float const lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 31));
        float const hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 63));
        return (threadIdx.x & 32u) ? hi : lo;
    }
}

// five candidates in two v_max3_f32 (the tree max(max(a, b), max(c, d)) has the shorter chain but is three
// instructions, and the row is bound by VALU issue -- a v_max* costs 4.2 SIMD cycles on gfx950 whatever its
// operand count: profiles/r03/valu_issue.txt); max is exact, so the grouping changes no result
// (written as asm: left to itself the compiler rebalances the chain of fmaxf into that tree here)
__device__ __forceinline__ float vmax3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max5(float a, float b, float c, float d, float e)
{
    return vmax3(vmax3(a, b, c), d, e);
}

// ---- vector loads of R consecutive floats ----------------------------------
template <int R> struct VecLoad;
template <> struct VecLoad<1>
{
    static __device__ __forceinline__ void ld(float const *p, float (&o)[1]) { o[0] = *p; }
};
template <> struct VecLoad<2>
{
    static __device__ __forceinline__ void ld(float const *p, float (&o)[2])
    {
        float2 v = *reinterpret_cast<float2 const *>(p);
        o[0] = v.x, o[1] = v.y;
    }
};
template <> struct VecLoad<3>
{
    static __device__ __forceinline__ void ld(float const *p, float (&o)[3])
    {
        // 12-byte rows: dwordx3 needs only dword alignment
        struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };
        f3 v = *reinterpret_cast<f3 const *>(p);
        o[0] = v.x, o[1] = v.y, o[2] = v.z;
    }
};
template <> struct VecLoad<4>
{
    static __device__ __forceinline__ void ld(float const *p, float (&o)[4])
    {
        float4 v = *reinterpret_cast<float4 const *>(p);
        o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
    }
};

template <> struct VecLoad<5>
{
    static __device__ __forceinline__ void ld(float const *p, float (&o)[5])
    {
        float t[3], u[2];
        VecLoad<3>::ld(p, t);
        struct __attribute__((packed, aligned(4))) f2 { float x, y; };
        f2 v = *reinterpret_cast<f2 const *>(p + 3);
        u[0] = v.x, u[1] = v.y;
        o[0] = t[0], o[1] = t[1], o[2] = t[2], o[3] = u[0], o[4] = u[1];
    }
};
template <> struct VecLoad<6>
{
    static __device__ __forceinline__ void ld(float const *p, float (&o)[6])
    {
        float t[3], u[3];
        VecLoad<3>::ld(p, t);
        VecLoad<3>::ld(p + 3, u);
        o[0] = t[0], o[1] = t[1], o[2] = t[2], o[3] = u[0], o[4] = u[1], o[5] = u[2];
    }
};
template <> struct VecLoad<7>
{
    static __device__ __forceinline__ void ld(float const *p, float (&o)[7])
    {
        struct __attribute__((packed, aligned(4))) f4 { float x, y, z, w; };
        f4 v = *reinterpret_cast<f4 const *>(p);
        float u[3];
        VecLoad<3>::ld(p + 4, u);
        o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w, o[4] = u[0], o[5] = u[1], o[6] = u[2];
    }
};
template <> struct VecLoad<8>
{
    static __device__ __forceinline__ void ld(float const *p, float (&o)[8])
    {
        float4 v = *reinterpret_cast<float4 const *>(p);
        float4 w = *reinterpret_cast<float4 const *>(p + 4);
        o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w, o[4] = w.x, o[5] = w.y, o[6] = w.z, o[7] = w.w;
    }
};

__device__ __forceinline__ unsigned code_of(unsigned w, int l)
{
    // offsets 0,4,20,84,340 for lengths 1..5; last base = least significant
    constexpr unsigned off[5] = {0u, 4u, 20u, 84u, 340u};
    return off[l - 1] + (w & ((1u << (2 * l)) - 1u));
}

// ---- per-pair DP state held in registers -----------------------------------
// The four emitting special states -- N, J, C of the alt model and R of the null model -- share one
// recursion shape, X(j) = max_l (PX(j-l) + e_bg(x[j-l..j))), PX(j) = max(E(j) + a, X(j) + b).  Lane t
// runs it for special t & 3 (0 N, 1 J, 2 C, 3 R) with its own a / b in registers, so a row pays for
// one such recursion instead of four; B(j) is put together inside each group of four lanes.
template <int R> struct PairState
{
    float P[5][R]; // P_k(j') = best predecessor of M_k leaving row j', slot j' % 5
    float Q[5][R]; // Q_k(j') = best predecessor of I_k
    float PX[5];   // same for this lane's special state
};

struct LaneSpecial
{
    float a; // E -> X:      -inf, EJ, EC, -inf
    float b; // X -> X:      NN, JJ, CC, RR
    float c; // X -> B:      NB, JB, -inf, -inf
};

template <int R> struct Trans
{
    float ent[R], mm[R], im[R], dm[R], md[R], dd[R], mi[R], ii[R];
};

struct RowOut
{
    float E; // wave-uniform
    float X; // this lane's special: N, J, C or R of the row
};

// Cross-wavefront exchange area of one block (W > 1 only).  Slots are indexed by
// a running barrier counter so a fast wave never overwrites what a slow wave is
// still reading: data written before barrier g lives in buffer g & 1 and is read
// only between barriers g and g+1; flags rotate over three words.
template <int W> struct Exchange
{
    float m[2][W], i[2][W], d[2][W], e[2][W];
    int flag[3];
};

// D of a lane's nodes 1..R-1 from its node 0 (the sequential part of the delete chain)
template <int R, int r = 1>
__device__ __forceinline__ void chain_rest(float const (&a)[R], float (&d)[R], float const (&dd)[R])
{
    if constexpr (r < R)
    {
        d[r] = fmaxf(a[r], d[r - 1] + dd[r]);
        chain_rest<R, r + 1>(a, d, dd);
    }
}

// One DP row. PH = j % 5 is compile-time so the history ring needs no moves.
// W == 1: the wavefront owns the whole profile. W > 1: wavefront `wave` owns
// nodes [wave*64*R, (wave+1)*64*R) and exchanges boundary values through LDS.
// `fetch` loads the NEXT row's emissions into em / eN / eI; it is called as soon as this row
// has consumed them (one set of registers, no copies), and its loads land during the
// cross-lane part of the row.
// PARTS > 1 (W == 1 only): the wavefront holds PARTS profiles side by side, 64 / PARTS lanes each (viterbi_mp_kernel):
// E(j) is then a per-part maximum, a value per lane, and eN / eI arrive per lane; everything else is as for one
// profile -- the edges into a profile's first node are -inf, so what the lane-shifts carry over from the neighbouring
// part's last lane never counts.
// `sink` (traceback only, viterbi_rowsweep_kernel<..., TRACE>): called once per row with the row's M, I, D of this lane's
// nodes, E(j), this lane's special and B(j); the scoring kernels pass nothing.
struct NoSink
{
    template <class... A> __device__ __forceinline__ void operator()(A const &...) const {}
};
template <int R, int W, int PH, int PARTS = 1, class Fetch, class Sink = NoSink>
__device__ __forceinline__ RowOut dp_row(PairState<R> &s, Trans<R> const &t,
                                         float (&em)[5][R], float (&em3)[R], float (&em4)[R], float (&em5)[R], float (&eN)[5], float (&eI)[5],
                                         LaneSpecial const &sp, float const xEB,
                                         Exchange<(W > 1 ? W : 1)> *xc,
                                         unsigned wave, unsigned lane,
                                         unsigned &gen, unsigned const exact_e, Fetch &&fetch, Sink &&sink = Sink{})
{
    constexpr int s1 = (PH + 4) % 5, s2 = (PH + 3) % 5, s3 = (PH + 2) % 5,
                  s4 = (PH + 1) % 5, s5 = PH;
    float const ni = neg_inf();

    // emitting states: value = max_l ( predecessor(j-l) + emission(x[j-l..j)) )
    float m[R], ins[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
        m[r] = max5(s.P[s1][r] + em[0][r], s.P[s2][r] + em[1][r],
                    s.P[s3][r] + em3[r], s.P[s4][r] + em4[r],
                    s.P[s5][r] + em5[r]);
        ins[r] = max5(s.Q[s1][r] + eI[0], s.Q[s2][r] + eI[1], s.Q[s3][r] + eI[2],
                      s.Q[s4][r] + eI[3], s.Q[s5][r] + eI[4]);
    }
    float const X = max5(s.PX[s1] + eN[0], s.PX[s2] + eN[1], s.PX[s3] + eN[2],
                         s.PX[s4] + eN[3], s.PX[s5] + eN[4]);
    __builtin_amdgcn_sched_barrier(0);
    fetch();
    __builtin_amdgcn_sched_barrier(0);

    // E(j), one wavefront per pair: the maximum over the MATCH states, taken before the delete chain (its seven
    // cross-lane steps no longer wait for the chain's fixed point).  With MD, DD <= 0 -- every profile whose
    // transitions are log-probabilities -- D_k <= max_{i<k} M_i (an add of a non-positive number never rounds
    // up), so the delete states cannot decide the maximum; a profile flagged DCP_PROF_EXACT_E at upload adds
    // them below (wave-uniform branch).  max is exact, so the order of its operands changes nothing.
    float E = ni;
    if constexpr (W == 1)
    {
        float em = m[0];
#pragma unroll
        for (int r = 1; r < R; ++r)
            em = fmaxf(em, m[r]);
        if constexpr (PARTS == 1) E = wave_max(em);
        else E = part_max<PARTS>(em);
    }

    // Delete chain D_k = max(M_{k-1} + MD_k, D_{k-1} + DD_k): sequential inside
    // a lane; across lanes (and wavefronts) iterate to the fixed point, which
    // is the sequential recurrence's unique solution -- exact, no reassociation.
    // A pass starts with the first node only: when no lane's D changes there (the usual case
    // after the first pass), the rest of the lane's chain cannot change either.
    // Node k-1 of a lane's first node sits in lane - 1 (W > 1: of lane 0, in the previous wavefront: *_first).
    float m_first = ni, i_first = ni, d_first = ni;
    constexpr bool XW = W > 1; // lane 0 may have a predecessor
    float a[R], d[R];
#pragma unroll
    for (int r = 1; r < R; ++r)
        a[r] = m[r - 1] + t.md[r];
    auto refine = [&]() {
        if constexpr (W == 1)
        {
            // D only grows from pass to pass, so the first node's new value is max(old, D of lane - 1 + DD) and
            // it changed exactly where that candidate exceeds the old value: test, then update in place
            // (no second register for the new value, no copy at the loop's back edge)
            for (;;)
            {
                float const c0 = shr1_add<XW>(d[R - 1], d_first, t.dd[0]);
                // compare FIRST, then the in-place maximum (as asm: the scheduler puts the maximum first,
                // which costs a second register and a copy per pass); all 64 lanes are active
                unsigned long long changed;
                asm("v_cmp_gt_f32_e64 %0, %2, %1\n\t"
                    "v_max_f32_e32 %1, %1, %2"
                    : "=&s"(changed), "+v"(d[0])
                    : "v"(c0));
                if (changed == 0ull) break;
                chain_rest<R>(a, d, t.dd);
            }
        }
        else
        {
            a[0] = shr1_add<XW>(m[R - 1], m_first, t.md[0]);
            for (;;)
            {
                float const d0 = fmaxf(a[0], shr1_add<XW>(d[R - 1], d_first, t.dd[0]));
                if (!__any(d0 != d[0])) break;
                d[0] = d0;
                chain_rest<R>(a, d, t.dd);
            }
        }
    };
    auto lane_max = [&]() {
        float e = fmaxf(m[0], d[0]);
#pragma unroll
        for (int r = 1; r < R; ++r)
            e = fmaxf(e, fmaxf(m[r], d[r]));
        return e;
    };
    // first pass: every lane's chain from its own M values (left neighbour's D not known yet)
    d[0] = shr1_add<XW>(m[R - 1], m_first, t.md[0]);
    chain_rest<R>(a, d, t.dd);
    refine();

    // E = max over nodes of M_k and D_k (exit scores are 0: protein_model.c:441-458)
    if constexpr (W == 1)
    {
        // (opaque: hoisted out of the row loops the condition lives as a lane mask and every row re-derives it
        // through a v_cndmask / v_cmp pair; this way it is one s_cmp + s_cbranch)
        unsigned xe = exact_e;
        asm volatile("" : "+s"(xe));
        if (__builtin_expect(xe != 0u, 0))
        {
            float ed = d[0];
#pragma unroll
            for (int r = 1; r < R; ++r)
                ed = fmaxf(ed, d[r]);
            // (back into an SGPR: merged with the common path as a VALU result, E would live in a VGPR and
            // the common path would pay a v_mov per row)
            E = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fmaxf(E, wave_max(ed)))));
        }
    }
    else
    {
        float published = ni; // lane 63: the D value the next wave last saw
        // this wavefront's part of E(j): the match states' maximum, once per row (as for W == 1: the delete
        // states cannot decide it unless the profile is flagged, and then they are added in every pass)
        float e_m = m[0];
#pragma unroll
        for (int r = 1; r < R; ++r)
            e_m = fmaxf(e_m, m[r]);
        e_m = wave_max(e_m);
        for (unsigned it = 0;; ++it, ++gen)
        {
            unsigned const buf = gen & 1u;
            float const e_wave = exact_e ? wave_max(lane_max()) : e_m;
            if (lane == 63)
            {
                if (it == 0)
                {
                    xc->m[buf][wave] = m[R - 1];
                    xc->i[buf][wave] = ins[R - 1];
                }
                else if (d[R - 1] != published)
                    xc->flag[gen % 3u] = 1;
                xc->d[buf][wave] = d[R - 1];
                xc->e[buf][wave] = e_wave;
                published = d[R - 1];
            }
            if (threadIdx.x == 0) xc->flag[(gen + 1u) % 3u] = 0;
            __syncthreads();
            if (it > 0 && xc->flag[gen % 3u] == 0)
            {
                E = xc->e[buf][0];
#pragma unroll
                for (int w = 1; w < W; ++w)
                    E = fmaxf(E, xc->e[buf][w]);
                ++gen;
                break;
            }
            if (wave > 0)
            {
                if (it == 0)
                {
                    m_first = xc->m[buf][wave - 1];
                    i_first = xc->i[buf][wave - 1];
                }
                d_first = xc->d[buf][wave - 1];
            }
            refine();
        }
    }

    // B(j) = max(N + NB, E + EB, J + JB)   (S(j>0) = -inf): lanes t & 3 = 0, 1 hold N, J
    float const B = fmaxf(quad_max(X + sp.c), E + xEB);
    sink(m, ins, d, E, X, B);

    // predecessors leaving this row (overwrite the slot of row j-5)
    s.P[PH][0] = fmaxf(fmaxf(B + t.ent[0], shr1_add<XW>(m[R - 1], m_first, t.mm[0])),
                       fmaxf(shr1_add<XW>(ins[R - 1], i_first, t.im[0]), shr1_add<XW>(d[R - 1], d_first, t.dm[0])));
    s.Q[PH][0] = fmaxf(m[0] + t.mi[0], ins[0] + t.ii[0]);
#pragma unroll
    for (int r = 1; r < R; ++r)
    {
        s.P[PH][r] = fmaxf(fmaxf(B + t.ent[r], m[r - 1] + t.mm[r]),
                           fmaxf(ins[r - 1] + t.im[r], d[r - 1] + t.dm[r]));
        s.Q[PH][r] = fmaxf(m[r] + t.mi[r], ins[r] + t.ii[r]);
    }
    s.PX[PH] = fmaxf(E + sp.a, X + sp.b);
    return RowOut{E, X};
}

// ---- segmented sweep (profiles of more than 512 nodes, one wavefront per pair) ------------------------------------
// One DP row of ONE SEGMENT of 64 x R consecutive nodes.  The wavefront sweeps the segments of a profile one after
// the other over all rows, the way the query-lane kernels sweep tiles: what row j of the next segment needs from
// this one -- M, I, D of the segment's last node and the running E -- goes through a per-wavefront scratch column
// (16 bytes per row), and B(j) is taken as N(j) + NB only; the last segment, which has the final E(j) and J(j),
// checks whether max(E(j) + EB, J(j) + JB) ever exceeded it.  If never, N + NB IS B (a forward recurrence has one
// solution) and the scores are exactly the multi-wavefront kernel's; a pair where it did goes to that kernel
// through a redo list.  `bin`: this row's boundary values from the previous segment (m, i, d of its last node, E so
// far; -inf for the first segment); `bout`: where lane 63 puts this segment's.
struct SegBnd
{
    float m, i, d, e;
};
template <int R, int PH, bool LASTSEG, class Fetch>
__device__ __forceinline__ RowOut seg_row(PairState<R> &s, Trans<R> const &t, float (&em)[5][R], float (&eN)[5],
                                          float (&eI)[5], LaneSpecial const &sp, float const cJ, float const xEB,
                                          SegBnd const &bin, float4 *bout, unsigned lane, bool &dirty, Fetch &&fetch)
{
    constexpr int s1 = (PH + 4) % 5, s2 = (PH + 3) % 5, s3 = (PH + 2) % 5, s4 = (PH + 1) % 5, s5 = PH;
    float m[R], ins[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
        m[r] = max5(s.P[s1][r] + em[0][r], s.P[s2][r] + em[1][r], s.P[s3][r] + em[2][r], s.P[s4][r] + em[3][r],
                    s.P[s5][r] + em[4][r]);
        ins[r] = max5(s.Q[s1][r] + eI[0], s.Q[s2][r] + eI[1], s.Q[s3][r] + eI[2], s.Q[s4][r] + eI[3],
                      s.Q[s5][r] + eI[4]);
    }
    float const X = max5(s.PX[s1] + eN[0], s.PX[s2] + eN[1], s.PX[s3] + eN[2], s.PX[s4] + eN[3], s.PX[s5] + eN[4]);
    __builtin_amdgcn_sched_barrier(0);
    fetch();
    __builtin_amdgcn_sched_barrier(0);

    // E(j) so far: the previous segments' and this one's match states (flagged profiles never come here)
    float em_ = m[0];
#pragma unroll
    for (int r = 1; r < R; ++r)
        em_ = fmaxf(em_, m[r]);
    float const E = fmaxf(bin.e, wave_max(em_));

    // delete chain: lane 0 continues the previous segment's chain (its D is final there)
    float a[R], d[R];
#pragma unroll
    for (int r = 1; r < R; ++r)
        a[r] = m[r - 1] + t.md[r];
    // (lane 0: max(M_prev + MD, D_prev + DD), both final; the other lanes' D(lane - 1) comes with the passes below)
    {
        float const first_only = shr1_add<true>(m[R - 1], bin.m, t.md[0]);
        d[0] = lane == 0u ? fmaxf(first_only, bin.d + t.dd[0]) : first_only;
    }
    chain_rest<R>(a, d, t.dd);
    for (;;)
    {
        float const c0 = shr1_add<true>(d[R - 1], bin.d, t.dd[0]);
        unsigned long long changed;
        asm("v_cmp_gt_f32_e64 %0, %2, %1\n\t"
            "v_max_f32_e32 %1, %1, %2"
            : "=&s"(changed), "+v"(d[0])
            : "v"(c0));
        if (changed == 0ull) break;
        chain_rest<R>(a, d, t.dd);
    }

    // what the next segment needs from this row
    if constexpr (!LASTSEG)
        if (lane == 63u) *bout = float4{m[R - 1], ins[R - 1], d[R - 1], E};

    // B(j) as this sweep takes it: N(j) + NB (lanes t & 3 = 0 hold N; cJ keeps J out)
    float const B = quad_max(X + sp.c);
    if constexpr (LASTSEG)
    {
        // did E(j) -> B(j) or J(j) -> B(j) beat it?  (X + cJ: J + JB in the lanes that hold J, -inf elsewhere)
        float const B1 = fmaxf(quad_max(X + cJ), E + xEB);
        dirty = dirty || B1 > B;
    }
    s.P[PH][0] = fmaxf(fmaxf(B + t.ent[0], shr1_add<true>(m[R - 1], bin.m, t.mm[0])),
                       fmaxf(shr1_add<true>(ins[R - 1], bin.i, t.im[0]), shr1_add<true>(d[R - 1], bin.d, t.dm[0])));
    s.Q[PH][0] = fmaxf(m[0] + t.mi[0], ins[0] + t.ii[0]);
#pragma unroll
    for (int r = 1; r < R; ++r)
    {
        s.P[PH][r] = fmaxf(fmaxf(B + t.ent[r], m[r - 1] + t.mm[r]), fmaxf(ins[r - 1] + t.im[r], d[r - 1] + t.dm[r]));
        s.Q[PH][r] = fmaxf(m[r] + t.mi[r], ins[r] + t.ii[r]);
    }
    s.PX[PH] = fmaxf(E + sp.a, X + sp.b);
    return RowOut{E, X};
}

// base `pos` of the sequence (a scalar load: the address is wave-uniform)
__device__ __forceinline__ unsigned base_at(cu32 *words, unsigned pos)
{
    return (words[pos >> 4] >> ((pos & 15u) * 2u)) & 3u;
}

__device__ __forceinline__ unsigned base_at(uint32_t const *__restrict__ words, unsigned pos)
{
    return (words[pos >> 4] >> ((pos & 15u) * 2u)) & 3u;
}

// The emission rows of one DP row: row `c` of the profile's table starts at a wave-uniform address
// (SGPR pair, pinned there by the empty asm) and every lane adds its own 32-bit byte offset -- the
// global_load form with an SGPR base, no per-lane 64-bit address arithmetic.  (`lane_boff` passes
// through an empty asm so that its zero-extension stays next to the load: hoisted out of the row
// loop as a 64-bit value it no longer matches that addressing mode.)
typedef char const __attribute__((address_space(1))) *gchar_ptr;
// STAGED > 0: the table's first STAGED rows -- the 4 + 16 rows of the one- and two-base words, which two of a
// DP row's five reads go to -- sit in the block's LDS (`stg`, same [code][ldk] layout): they are the rows
// every query of the profile keeps re-reading, and out of LDS they cost the L1 / L2 path nothing.
typedef float const __attribute__((address_space(3))) *lds_cfloat_ptr;
// LMASK: bit l - 1 set = load the rows of the l-base word (the rows of three to five bases go to em3 / em4 / em5, which a
// caller prefetching them two DP rows ahead points at the buffer of that row's parity); SCALARS: also the
// insert / background emissions of the row.
template <int R, int STAGED, int LMASK = 31, bool SCALARS = true>
__device__ __forceinline__ void load_row(float const *__restrict__ em_base,
                                         unsigned ldk, unsigned &lane_boff,
                                         cfloat *eN_tab, cfloat *eI_tab,
                                         unsigned w, float (&em)[5][R], float (&em3)[R], float (&em4)[R], float (&em5)[R],
                                         float (&eN)[5], float (&eI)[5], float const *stg, unsigned stg_ldk = 0u)
{
    // stg_ldk: row length of the staged image when it is not the table's (a segment's columns only); 0 = ldk
    unsigned const sldk = stg_ldk ? stg_ldk : ldk;
    asm volatile("" : "+v"(lane_boff));
#pragma unroll
    for (int l = 1; l <= 5; ++l)
    {
        unsigned const c = code_of(w, l);
        if ((LMASK >> (l - 1)) & 1)
        {
            float(&dst)[R] = l == 3 ? em3 : l == 4 ? em4 : l == 5 ? em5 : em[l - 1];
            if (STAGED > 0 && (l == 1 ? 4 : l == 2 ? 20 : l == 3 ? 84 : 1364) <= STAGED) // the staged rows hold all words of l bases
            {
                // the row's byte offset as ONE scalar value (left alone, the constant part of a two-base word's
                // row becomes a second per-lane add: ds_read2's offset fields are too narrow for it)
                unsigned roff = c * sldk * 4u;
                asm volatile("" : "+s"(roff));
                VecLoad<R>::ld((float const *)((char const *)stg + (roff + lane_boff)), dst);
            }
            else
            {
                gchar_ptr row = (gchar_ptr)em_base + c * ldk * 4u; // < 2^32: 1364 codes x 4096 nodes x 4 B
                asm volatile("" : "+s"(row));
                VecLoad<R>::ld((float const *)(row + lane_boff), dst);
            }
        }
        if (SCALARS)
        {
            eN[l - 1] = eN_tab[c];
            eI[l - 1] = eI_tab[c];
        }
    }
}

} // namespace

// ============================================================================
// Exact row-sweep Viterbi.  Lane t of wavefront w owns nodes
// [(w*64+t)*R, (w*64+t)*R + R) of the profile (core_size <= 64*R*W).
//   W == 1: block = 4 independent wavefronts, each its own (profile, query
//           chunk) task, consecutive chunks of the same profile so that their
//           emission-table reads share one XCD's L2;
//   W  > 1: block = W cooperating wavefronts on one task.
// ============================================================================
// wavefronts per SIMD the register allocation must leave room for (512 VGPRs per SIMD lane)
#ifndef DCP_RS_R4_WAVES
#define DCP_RS_R4_WAVES 4 // wavefronts per SIMD of the R = 4 one-wavefront class: 4 = 128 VGPRs with 23 spilled (35 GB of scratch traffic per C3 launch), 3 = 168 and none -- measured: 538 vs 573 ms, occupancy wins
#endif
#ifndef DCP_RS_R2_WAVES
#define DCP_RS_R2_WAVES 6 // wavefronts per SIMD of the R = 2 class: 6 = 80 VGPRs with 5 spilled, against 5 at 85 without --
#endif                     // R2W1 launch 712 -> 683 ms (7 = 72 VGPRs, 27 spilled: another 1.4 %, not taken)
// R = 5 and 6: three wavefronts per SIMD (168 VGPRs, 53 / 173 spilled) instead of two: R5W1 launch 448 -> 384 ms,
// R6W1 269 -> 228; R = 7 at three (425 spilled: 171 -> 484 ms) and R = 5 at four (316 spilled: 374 -> 673) lose
#ifndef DCP_RS_R5_WAVES
#define DCP_RS_R5_WAVES 3
#endif
#ifndef DCP_RS_R6_WAVES
#define DCP_RS_R6_WAVES 3
#endif
#ifndef DCP_RS_R7_WAVES
#define DCP_RS_R7_WAVES 2
#endif
#ifndef DCP_RS_R4_BIG_WAVES
#define DCP_RS_R4_BIG_WAVES DCP_RS_R4_WAVES
#endif
#ifndef DCP_RS_PAIR_BIG
#define DCP_RS_PAIR_BIG 0 // 1: the pair-mode kernels (redo lists) at the large-batch variants' occupancy
#endif
#ifndef DCP_RS_R2_BIG_WAVES
#define DCP_RS_R2_BIG_WAVES 7 // R2W1 launch 694 -> 684 ms (72 VGPRs, 27 spilled)
#endif
#ifndef DCP_RS_R3_BIG_WAVES
#define DCP_RS_R3_BIG_WAVES 5 // R3W1 launch 754 -> 739 ms (96 VGPRs)
#endif
// `big`: the variant of the large batches (20 rows staged, one row of prefetch).  Only there do R = 5, 6 run three
// wavefronts per SIMD: the small batches wait for HBM, and the spilled registers' scratch traffic costs them more
// than the third wavefront hides (1 query 11.7 -> 13.9 ms, 16 queries 80 -> 84 with it everywhere).
constexpr int rs_min_waves(int R, int W = 1, bool big = false)
{
    return R == 2 ? (big ? DCP_RS_R2_BIG_WAVES : DCP_RS_R2_WAVES) : R == 3 && W == 1 && big ? DCP_RS_R3_BIG_WAVES : R <= 3 ? 4
           : R == 4 ? (W == 1 ? (big ? DCP_RS_R4_BIG_WAVES : DCP_RS_R4_WAVES) : 4)
           : R == 5 ? (big ? DCP_RS_R5_WAVES : 2) : R == 6 ? (big ? DCP_RS_R6_WAVES : 2) : R == 7 ? DCP_RS_R7_WAVES : 2;
}
constexpr int rs_block_threads(int R, int W, int STG, bool PF = false)
{
    return W > 1 ? 64 * W : STG > 0 ? (rs_min_waves(R, W, STG == 20 && !PF) >= 4 ? 1024 : 256 * rs_min_waves(R, W, STG == 20 && !PF)) : 256;
}
// PF (staged variants): the rows that still come from global memory are fetched TWO DP rows ahead -- the variants of
// the small batches, which wait for HBM latency (one query: a wavefront's row takes as long as its loads).
// TRACE (pair mode, unstaged): the forward half of the traceback -- the same rows, and every row's M, I, D of every node and
// its N, B, E, J, C go to the pair's work area (dcp_scan_args::trace_work + trace_woff[task]: three matrices [L + 1][64 R W]
// and five vectors [L + 1]); no score or hit record is written, the pair's alt score goes to trace_alt[task].
template <int R, int W, int STG, bool PF, bool TRACE = false>
__global__ __launch_bounds__(rs_block_threads(R, W, STG, PF), TRACE ? (W > 4 ? 1 : 2) : rs_min_waves(R, W, (STG == 20 && !PF) || (STG == 0 && W == 1 && DCP_RS_PAIR_BIG))) void viterbi_rowsweep_kernel(dcp_scan_args a)
{
    static_assert(!PF || STG > 0, "only staged variants prefetch two rows ahead");
    static_assert(!TRACE || (STG == 0 && !PF), "the traceback's forward pass is an unstaged pair-mode kernel");
    static_assert(W == 1 || STG == 0, "only one-wavefront pairs stage rows");
    constexpr unsigned TASKS_PER_BLOCK = (W == 1 && STG == 0) ? 4u : 1u;
    constexpr int STAGED = STG;
    constexpr bool PF2 = PF;
    constexpr int NEAR = STG >= 84 ? 7 : 3, FAR = 31 - NEAR; // word lengths read from LDS / from global memory
    __shared__ Exchange<(W > 1 ? W : 1)> xc_mem;
    // (R <= 2: a profile that shares its rows with others owns core_size + 8 columns, up to 8 more than 64 x R)
    __shared__ __attribute__((aligned(16))) float stage_mem[STAGED > 0 ? STAGED * (64 * R + (R <= 2 ? 8 : 0)) : 4];
    float const *const stg = stage_mem;
    Exchange<(W > 1 ? W : 1)> *xc = &xc_mem;
    unsigned const lane = threadIdx.x & 63u;
    unsigned const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // XCD-aware task map: blocks with equal (blockIdx % 8) share an XCD/L2, so
    // give each XCD one contiguous range of tasks (= consecutive chunks of the
    // same few profiles).  Placement only affects speed, never results.
    unsigned const nblk = gridDim.x; // multiple of 8
    unsigned const vblk = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    bool const pair_mode = STAGED > 0 ? false : a.pairs != nullptr;
    // staged: block -> (profile, group of `bw` chunks); a profile's chunk count is rounded up to whole blocks, so
    // the wavefronts of a block always share the profile (the last block's spare wavefronts only help with the copy)
    unsigned const bw = STAGED > 0 ? blockDim.x >> 6 : 1u;
    unsigned const bpp = STAGED > 0 ? (a.nchunks + bw - 1u) / bw : 1u; // blocks per profile
    unsigned task = TASKS_PER_BLOCK > 1u ? vblk * TASKS_PER_BLOCK + wave : vblk;
    unsigned const ntasks = pair_mode ? min(*a.npairs, a.pair_cap) : STAGED > 0 ? a.nprof * bpp : a.nprof * a.nchunks;
    for (; task < ntasks; task += nblk * TASKS_PER_BLOCK) // uniform per task: whole block for W > 1 and staged blocks
    {
    unsigned slot, q0, q1;
    if (pair_mode)
    {
        dcp_pair const pr = a.pairs[task];
        slot = pr.slot, q0 = pr.q, q1 = pr.q + 1u;
    }
    else if constexpr (STAGED > 0)
    {
        unsigned const s_rel = task / bpp;
        unsigned const chunk = (task - s_rel * bpp) * bw + wave;
        slot = a.first_prof + s_rel;
        q0 = min(chunk * a.qchunk, a.nseqs); // a spare wavefront: q0 == q1
        q1 = min(q0 + a.qchunk, a.nseqs);
    }
    else
    {
        unsigned const s_rel = task / a.nchunks;
        slot = a.first_prof + s_rel;
        q0 = (task - s_rel * a.nchunks) * a.qchunk;
        q1 = min(q0 + a.qchunk, a.nseqs);
    }
    // the task is the same for every lane of the wavefront: say so, and everything derived from
    // it (table bases, sequence words, window codes) stays in SGPRs
    slot = __builtin_amdgcn_readfirstlane(slot);
    q0 = __builtin_amdgcn_readfirstlane(q0);
    q1 = __builtin_amdgcn_readfirstlane(q1);

    dcp_prof_meta const pm = a.profs[slot];
    float const *__restrict__ em_base = a.emis_match + pm.emis_off;
    cfloat *eN_tab = as_const(a.emis_null + (size_t)pm.pidx * DCP_NCODES);
    cfloat *eI_tab = as_const(a.emis_insert + (size_t)pm.pidx * DCP_NCODES);
    // Row length of the profile's tables: core_size + R (+ 8 for the classes of several wavefronts) rounded up to
    // 4, at most the class capacity, so that every row (emissions and transitions alike) ends in columns of -inf:
    // a lane past the last node reads THOSE -- the same few bytes for all such lanes -- instead of owning padding
    // columns of its own, and a row is 15-20 % shorter on a Pfam-like size distribution (dcp_gpu_db_upload).
    unsigned const ldk = pm.ldk;
    // wave-uniform and opaque: a scalar compare and branch per row (as a plain bool the condition is
    // re-materialised through a v_cndmask / v_cmp pair in every row)
    unsigned const exact_e = __builtin_amdgcn_readfirstlane(pm.flags & DCP_PROF_EXACT_E);
    unsigned const node0 = (W == 1 ? lane : wave * 64u + lane) * R; // this lane's first node
    // past the last node: R columns of the -inf padding behind the profile's own -- the first aligned ones if they
    // fit (always, when the profile shares its rows with others: ldk is then the shared row's length), else the row's end
    unsigned const m4 = (pm.core_size + 3u) & ~3u;
    unsigned const lane_off = node0 < pm.core_size ? node0 : (m4 + R <= pm.width ? m4 : pm.width - R);
    unsigned gen = 0;
    unsigned const swd = pm.width; // row length of the staged image: the profile's own columns
    if constexpr (STAGED > 0)
    {
        // the profile's columns of the table's first STAGED rows: [code][swd] in LDS (a stand-alone table's rows are
        // that long and the copy is contiguous; a profile that shares its rows with others has them ldk apart)
        unsigned const n4 = swd >> 2; // widths are multiples of 4
        for (unsigned i = threadIdx.x; i < (unsigned)STAGED * n4; i += blockDim.x)
        {
            unsigned const c = i / n4, k = i - c * n4;
            reinterpret_cast<float4 *>(stage_mem + c * swd)[k] = reinterpret_cast<float4 const *>(em_base + (size_t)c * ldk)[k];
        }
        __syncthreads();
    }
    if constexpr (W > 1)
    {
        __syncthreads(); // pair mode: the previous task's last exchange is over
        if (threadIdx.x == 0) xc->flag[0] = xc->flag[1] = xc->flag[2] = 0;
        __syncthreads();
    }

    Trans<R> t;
    {
        float const *__restrict__ tb = a.trans8 + pm.trans_off + lane_off;
        VecLoad<R>::ld(tb + (size_t)DCP_T_ENTRY * ldk, t.ent);
        VecLoad<R>::ld(tb + (size_t)DCP_T_MM * ldk, t.mm);
        VecLoad<R>::ld(tb + (size_t)DCP_T_IM * ldk, t.im);
        VecLoad<R>::ld(tb + (size_t)DCP_T_DM * ldk, t.dm);
        VecLoad<R>::ld(tb + (size_t)DCP_T_MD * ldk, t.md);
        VecLoad<R>::ld(tb + (size_t)DCP_T_DD * ldk, t.dd);
        VecLoad<R>::ld(tb + (size_t)DCP_T_MI * ldk, t.mi);
        VecLoad<R>::ld(tb + (size_t)DCP_T_II * ldk, t.ii);
    }

    for (unsigned q = q0; q < q1; ++q)
    {
        unsigned const L = a.seq_len[q];
        cu32 *words = as_const(a.seq_words + a.seq_woff[q]);
        cfloat *xt = as_const(a.xtrans + (size_t)q * DCP_XSTRIDE);
        float const ni = neg_inf();
        unsigned const x = lane & 3u; // this lane's special state: 0 N, 1 J, 2 C, 3 R
        LaneSpecial sp;
        sp.a = x == 1u ? xt[DCP_X_EJ] : x == 2u ? xt[DCP_X_EC] : ni;
        sp.b = x == 0u ? xt[DCP_X_NN] : x == 1u ? xt[DCP_X_JJ] : x == 2u ? xt[DCP_X_CC] : xt[DCP_X_RR];
        sp.c = x == 0u ? xt[DCP_X_NB] : x == 1u ? xt[DCP_X_JB] : ni;
        float xEB = xt[DCP_X_EB];
        asm volatile("" : "+v"(xEB)); // a VGPR: E(j) arrives in an SGPR, and an add takes only one

        // row 0: S = 0, B = S + SB, everything else -inf
        PairState<R> s;
#pragma unroll
        for (int h = 0; h < 5; ++h)
        {
#pragma unroll
            for (int r = 0; r < R; ++r)
                s.P[h][r] = ni, s.Q[h][r] = ni;
            s.PX[h] = ni;
        }
        {
            float const B0 = 0.0f + xt[DCP_X_SB];
#pragma unroll
            for (int r = 0; r < R; ++r)
                s.P[0][r] = B0 + t.ent[r];
            // PN(0) = S + SN; PR(0) = start lprob of R = 0 (protein_model.c:224)
            s.PX[0] = x == 0u ? 0.0f + xt[DCP_X_SN] : x == 3u ? 0.0f : ni;
        }

        float em[5][R], eN[5], eI[5];
        RowOut o{ni, ni};
        unsigned j = 1;
        unsigned lane_boff = lane_off * 4u; // pinned in place by load_row's asm (by value it is copied per row)
        // traceback: where this lane's values of the NEXT row go (row 0 is written here)
        constexpr unsigned RW = 64u * R * W;
        float *t_m = nullptr, *t_x = nullptr, *t_e = nullptr;
        size_t t_mat = 0;
        unsigned t_rows = 0;
        if constexpr (TRACE)
        {
            float *const work = a.trace_work + a.trace_woff[task];
            t_rows = L + 1u;
            t_mat = (size_t)t_rows * RW;
            t_m = work + node0;
            float *const spec = work + 3u * t_mat; // N, B, E, J, C: [L + 1] each
#pragma unroll
            for (int r = 0; r < R; ++r)
                t_m[r] = ni, t_m[t_mat + r] = ni, t_m[2u * t_mat + r] = ni;
            t_m += RW;
            // lanes 0, 1, 2 of the first wavefront hold N, J, C; lane 0 also writes B and E
            t_x = spec + (x == 0u ? 0u : x == 1u ? 3u * t_rows : 4u * t_rows);
            t_e = spec;
            if ((W == 1 || wave == 0u) && lane < 3u) t_x[0] = ni;
            if ((W == 1 || wave == 0u) && lane == 0u) t_e[t_rows] = 0.0f + xt[DCP_X_SB], t_e[2u * t_rows] = ni;
            ++t_x, ++t_e;
        }
        auto const sink = [&](float const(&m_)[R], float const(&i_)[R], float const(&d_)[R], float E_, float X_, float B_) {
            if constexpr (TRACE)
            {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    t_m[r] = m_[r], t_m[t_mat + r] = i_[r], t_m[2u * t_mat + r] = d_[r];
                t_m += RW;
                if ((W == 1 || wave == 0u) && lane < 3u) *t_x = X_;
                if ((W == 1 || wave == 0u) && lane == 0u) t_e[t_rows] = B_, t_e[2u * t_rows] = E_;
                ++t_x, ++t_e;
            }
        };
        if constexpr (PF2)
        {
            // The rows that still come from global memory (the four- and five-base words) are fetched TWO DP rows
            // ahead into the buffer of that row's parity: a small batch streams them from HBM and is short of
            // loads in flight, not of bytes (profiles/r03/rowsweep_variants.txt).  Rows are unrolled by ten (phase
            // j % 5 x parity); the words past the last base are padding.
            float emg[2][3][R]; // [parity][three-, four-, five-base rows] (the three-base slot is unused with 84 rows staged)
            unsigned w1 = base_at(words, 0);                                   // window of row 1
            unsigned w2 = ((w1 << 2) | base_at(words, 1)) & 1023u;              // window of row 2
            load_row<R, STAGED, NEAR, true>(em_base, ldk, lane_boff, eN_tab, eI_tab, w1, em, em[2], em[3], em[4], eN, eI, stg, swd);
            load_row<R, STAGED, FAR, false>(em_base, ldk, lane_boff, eN_tab, eI_tab, w1, em, emg[1][0], emg[1][1], emg[1][2], eN, eI, stg, swd);
            load_row<R, STAGED, FAR, false>(em_base, ldk, lane_boff, eN_tab, eI_tab, w2, em, emg[0][0], emg[0][1], emg[0][2], eN, eI, stg, swd);
#define DCP_E3(PAR) (STG >= 84 ? em[2] : emg[PAR][0])
#define DCP_ROW2(PH, PAR)                                                      \
    {                                                                          \
        w1 = w2;                                               /* row j + 1 */ \
        w2 = ((w2 << 2) | base_at(words, j + 1u)) & 1023u;     /* row j + 2 */ \
        o = dp_row<R, W, PH>(s, t, em, DCP_E3(PAR), emg[PAR][1], emg[PAR][2], eN, eI, sp, xEB, xc, wave, lane, gen, exact_e, [&]() { \
            load_row<R, STAGED, NEAR, true>(em_base, ldk, lane_boff, eN_tab, eI_tab, w1, em, em[2], em[3], em[4], eN, eI, stg, swd); \
            load_row<R, STAGED, FAR, false>(em_base, ldk, lane_boff, eN_tab, eI_tab, w2, em, emg[PAR][0], emg[PAR][1], emg[PAR][2], eN, eI, stg, swd); \
        });                                                                    \
        ++j;                                                                   \
    }
            while (j + 9 <= L)
            {
                DCP_ROW2(1, 1) DCP_ROW2(2, 0) DCP_ROW2(3, 1) DCP_ROW2(4, 0) DCP_ROW2(0, 1)
                DCP_ROW2(1, 0) DCP_ROW2(2, 1) DCP_ROW2(3, 0) DCP_ROW2(4, 1) DCP_ROW2(0, 0)
            }
            if (j <= L) DCP_ROW2(1, 1)
            if (j <= L) DCP_ROW2(2, 0)
            if (j <= L) DCP_ROW2(3, 1)
            if (j <= L) DCP_ROW2(4, 0)
            if (j <= L) DCP_ROW2(0, 1)
            if (j <= L) DCP_ROW2(1, 0)
            if (j <= L) DCP_ROW2(2, 1)
            if (j <= L) DCP_ROW2(3, 0)
            if (j <= L) DCP_ROW2(4, 1)
#undef DCP_ROW2
#undef DCP_E3
        }
        else
        {
        unsigned w = base_at(words, 0);
        load_row<R, STAGED>(em_base, ldk, lane_boff, eN_tab, eI_tab, w, em, em[2], em[3], em[4], eN, eI, stg, swd);

// compute row j from the tables in registers; once consumed they are refilled for row j+1
// (the word one past the last base is padding: harmless)
#define DCP_ROW(PH)                                                            \
    {                                                                          \
        w = ((w << 2) | base_at(words, j)) & 1023u;                            \
        o = dp_row<R, W, PH>(s, t, em, em[2], em[3], em[4], eN, eI, sp, xEB, xc, wave, lane, gen, exact_e, [&]() { \
            load_row<R, STAGED>(em_base, ldk, lane_boff, eN_tab, eI_tab, w, em, em[2], em[3], em[4], eN, eI, stg, swd); \
        }, sink);                                                              \
        ++j;                                                                   \
    }
        while (j + 4 <= L)
        {
            DCP_ROW(1) DCP_ROW(2) DCP_ROW(3) DCP_ROW(4) DCP_ROW(0)
        }
        if (j <= L) DCP_ROW(1)
        if (j <= L) DCP_ROW(2)
        if (j <= L) DCP_ROW(3)
        if (j <= L) DCP_ROW(4)
        }
#undef DCP_ROW

        // C(L) and R(L) sit in lanes 2 and 3
        float const C = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o.X), 2));
        float const nul = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o.X), 3));
        float const alt = fmaxf(o.E + xt[DCP_X_ET], C + xt[DCP_X_CT]);
        if constexpr (TRACE)
        {
            if (threadIdx.x == (W == 1 ? wave * 64u : 0u)) a.trace_alt[task] = alt;
            continue;
        }
        if (threadIdx.x == (W == 1 ? wave * 64u : 0u))
        {
            size_t const oi = (size_t)q * a.nprof_total + pm.pidx;
            if (a.out_null) a.out_null[oi] = nul;
            if (a.out_alt) a.out_alt[oi] = alt;
            // xmath_lrt_f32 + filter of scan_thread.c:121-123
            float const lrt = -2 * (nul - alt);
            if (__builtin_isfinite(lrt) && !(lrt < a.lrt_threshold))
            {
                unsigned const h = atomicAdd(a.nhits, 1u);
                if (h < a.hit_cap)
                    a.hits[h] = dcp_hit{a.q_base + q, pm.pidx, nul, alt};
            }
        }
    }
    if (!pair_mode) break; // grid mode: one task per block
    }
}

// ============================================================================
// K profiles per wavefront (round 4): the classes of at most 64 / at most 128 nodes, grid mode.
// One node per lane (R = 1) pays a row's cross-lane work -- E(j)'s seven-step maximum, the delete chain's passes, the
// lane shifts -- for 45 cells on average, two per lane for 95: 594 and 931 Gcell/s against 1 100-1 300 for the wider
// classes.  Here K = 4 (at most 64 nodes) or 2 (at most 128) profiles share a wavefront, 64 / K lanes of FOUR nodes
// each, scored against the SAME query: one sequence window, one set of special transitions, one row count for all
// of them, a row's cross-lane work once for 180-190 cells.  The K profiles' tables lie side by side in one table
// (dcp_mp_group), so a row is still one scalar row base + the lane's byte offset; E(j) is a per-part maximum
// (part_max); the insert / background emissions come per lane from the group's [1364][K] float2 table.  A block is
// four wavefronts on four consecutive queries of one group, with the table's 20 leading rows in LDS.
// ============================================================================
template <int K>
__global__ __launch_bounds__(256, 3) void viterbi_mp_kernel(dcp_scan_args a) // (four wavefronts per SIMD at 128 VGPRs spill 44: 3 % slower)
{
    constexpr int R = 4;
    constexpr unsigned PL = 64u / K;       // lanes per profile
    constexpr unsigned MAXLD = 4u * 72u + 16u; // longest row of a group: 4 x (64 + 8) or 2 x (128 + 8) columns
    constexpr int STAGED = 20;
    __shared__ __attribute__((aligned(16))) float stage_mem[STAGED * MAXLD];
    unsigned const lane = threadIdx.x & 63u;
    unsigned const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned const nblk = gridDim.x; // multiple of 8
    unsigned const vblk = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3); // XCD-aware, as in the row sweep
    unsigned const bpp = (a.nseqs + 3u) / 4u; // blocks per group: four queries each
    if (vblk >= a.nprof * bpp) return;
    unsigned const g_rel = vblk / bpp;
    unsigned const qi = (vblk - g_rel * bpp) * 4u + wave;
    bool const has_q = qi < a.nseqs; // a spare wavefront only helps with the copy
    unsigned const q = __builtin_amdgcn_readfirstlane(has_q ? qi : 0u);
    dcp_mp_group const *gp = a.mp_groups + (a.first_prof + g_rel);
    unsigned const ldk = __builtin_amdgcn_readfirstlane(gp->ldk);
    unsigned const nparts = __builtin_amdgcn_readfirstlane(gp->nparts);
    float const *__restrict__ em_base = a.emis_match + gp->emis_off;
    {
        // the group's table is a stand-alone [1364][ldk]: its first STAGED rows are contiguous
        float4 const *__restrict__ src = reinterpret_cast<float4 const *>(em_base);
        float4 *dst = reinterpret_cast<float4 *>(stage_mem);
        for (unsigned i = threadIdx.x; i < (unsigned)STAGED * (ldk >> 2); i += 256u)
            dst[i] = src[i];
        __syncthreads();
    }
    if (!has_q) return;
    float const *const stg = stage_mem;
    float const ni = neg_inf();
    unsigned const part = lane / PL, pl = lane % PL;
    unsigned const M = gp->core_size[part];
    unsigned const node0 = pl * R;
    // past the profile's last node (or an absent member): the four aligned columns of -inf behind it
    unsigned const lane_off = gp->col0[part] + (node0 < M ? node0 : ((M + 3u) & ~3u));
    Trans<R> t;
    {
        float const *__restrict__ tb = a.trans8 + gp->trans_off + lane_off;
        VecLoad<R>::ld(tb + (size_t)DCP_T_ENTRY * ldk, t.ent);
        VecLoad<R>::ld(tb + (size_t)DCP_T_MM * ldk, t.mm);
        VecLoad<R>::ld(tb + (size_t)DCP_T_IM * ldk, t.im);
        VecLoad<R>::ld(tb + (size_t)DCP_T_DM * ldk, t.dm);
        VecLoad<R>::ld(tb + (size_t)DCP_T_MD * ldk, t.md);
        VecLoad<R>::ld(tb + (size_t)DCP_T_DD * ldk, t.dd);
        VecLoad<R>::ld(tb + (size_t)DCP_T_MI * ldk, t.mi);
        VecLoad<R>::ld(tb + (size_t)DCP_T_II * ldk, t.ii);
    }
    // {eI, eN} of word code c for this lane's profile: in_tab[c][part]
    float2 const *__restrict__ in_tab = reinterpret_cast<float2 const *>(a.mp_in) + gp->in_off;
    unsigned part_boff = part * 8u;

    unsigned const L = a.seq_len[q];
    cu32 *words = as_const(a.seq_words + a.seq_woff[q]);
    cfloat *xt = as_const(a.xtrans + (size_t)q * DCP_XSTRIDE);
    unsigned const x = lane & 3u; // this lane's special state: 0 N, 1 J, 2 C, 3 R
    LaneSpecial sp;
    sp.a = x == 1u ? xt[DCP_X_EJ] : x == 2u ? xt[DCP_X_EC] : ni;
    sp.b = x == 0u ? xt[DCP_X_NN] : x == 1u ? xt[DCP_X_JJ] : x == 2u ? xt[DCP_X_CC] : xt[DCP_X_RR];
    sp.c = x == 0u ? xt[DCP_X_NB] : x == 1u ? xt[DCP_X_JB] : ni;
    float const xEB = xt[DCP_X_EB];

    PairState<R> s;
#pragma unroll
    for (int h = 0; h < 5; ++h)
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
            s.P[h][r] = ni, s.Q[h][r] = ni;
        s.PX[h] = ni;
    }
    {
        float const B0 = 0.0f + xt[DCP_X_SB];
#pragma unroll
        for (int r = 0; r < R; ++r)
            s.P[0][r] = B0 + t.ent[r];
        s.PX[0] = x == 0u ? 0.0f + xt[DCP_X_SN] : x == 3u ? 0.0f : ni;
    }

    float em[5][R], eN[5], eI[5];
    RowOut o{ni, ni};
    unsigned j = 1;
    unsigned lane_boff = lane_off * 4u;
    unsigned gen = 0;
    // one row's inputs: the match emission rows (two from LDS, three from global memory: load_row) and the lane's
    // {eI, eN} pairs -- scalar row pointer + the lane's part offset, like the emission rows
    auto fetch_in = [&](unsigned w) {
        load_row<R, STAGED, 31, false>(em_base, ldk, lane_boff, nullptr, nullptr, w, em, em[2], em[3], em[4], eN, eI, stg);
        asm volatile("" : "+v"(part_boff));
#pragma unroll
        for (int l = 1; l <= 5; ++l)
        {
            gchar_ptr row = (gchar_ptr)in_tab + code_of(w, l) * (unsigned)K * 8u;
            asm volatile("" : "+s"(row));
            float2 const v = *(float2 const *)(row + part_boff);
            eI[l - 1] = v.x, eN[l - 1] = v.y;
        }
    };
    unsigned w = base_at(words, 0);
    fetch_in(w);
#define DCP_MROW(PH)                                                                                   \
    {                                                                                                 \
        w = ((w << 2) | base_at(words, j)) & 1023u;                                                   \
        o = dp_row<R, 1, PH, K>(s, t, em, em[2], em[3], em[4], eN, eI, sp, xEB, nullptr, wave, lane, gen, 0u, \
                                [&]() { fetch_in(w); });                                              \
        ++j;                                                                                          \
    }
    while (j + 4 <= L)
    {
        DCP_MROW(1) DCP_MROW(2) DCP_MROW(3) DCP_MROW(4) DCP_MROW(0)
    }
    if (j <= L) DCP_MROW(1)
    if (j <= L) DCP_MROW(2)
    if (j <= L) DCP_MROW(3)
    if (j <= L) DCP_MROW(4)
#undef DCP_MROW

    // C(L) and R(L) sit in lanes 2 and 3 of every quad: the first lane of each part takes its own quad's
    float const C = dpp_mov<0xAA /*quad_perm:[2,2,2,2]*/, 0xf, 0xf>(o.X, o.X);
    float const nul = dpp_mov<0xFF /*quad_perm:[3,3,3,3]*/, 0xf, 0xf>(o.X, o.X);
    float const alt = fmaxf(o.E + xt[DCP_X_ET], C + xt[DCP_X_CT]);
    if (pl == 0u && part < nparts)
    {
        unsigned const pidx = gp->pidx[part];
        size_t const oi = (size_t)q * a.nprof_total + pidx;
        if (a.out_null) a.out_null[oi] = nul;
        if (a.out_alt) a.out_alt[oi] = alt;
        // xmath_lrt_f32 + filter of scan_thread.c:121-123
        float const lrt = -2 * (nul - alt);
        if (__builtin_isfinite(lrt) && !(lrt < a.lrt_threshold))
        {
            unsigned const h = atomicAdd(a.nhits, 1u);
            if (h < a.hit_cap) a.hits[h] = dcp_hit{a.q_base + q, pidx, nul, alt};
        }
    }
}

// the groups' {insert, background} tables from the per-profile ones: out[(in_off + c * K + part)] = {eI, eN}
__global__ __launch_bounds__(256) void mp_in_kernel(dcp_mp_group const *groups, unsigned n_first, unsigned k_first, unsigned k_rest,
                                                    float const *emis_insert, float const *emis_null, float *out)
{
    dcp_mp_group const g = groups[blockIdx.x];
    unsigned const K = blockIdx.x < n_first ? k_first : k_rest;
    float2 *dst = reinterpret_cast<float2 *>(out) + g.in_off;
    for (unsigned i = threadIdx.x; i < (unsigned)DCP_NCODES * K; i += 256u)
    {
        unsigned const c = i / K, part = i - c * K;
        // an absent member: -inf (its lanes' values are never used, but must not be NaN sources)
        float2 v{neg_inf(), neg_inf()};
        if (part < g.nparts) v = float2{emis_insert[(size_t)g.pidx[part] * DCP_NCODES + c], emis_null[(size_t)g.pidx[part] * DCP_NCODES + c]};
        dst[i] = v;
    }
}

// ============================================================================
// Segmented sweep, SEGMENT-MAJOR (round 4): one wavefront per (profile, query) pair of a multi-wavefront size class,
// the profile cut into segments of 64 x R nodes (seg_row above) -- but ONE SEGMENT PER LAUNCH: launch s sweeps segment
// s of every pair of the class over all rows, reading what segment s - 1 left for each row (M, I, D of its last node,
// E so far: 16 bytes) from the pair's boundary column in HBM and writing its own to the other column of the pair.
// Why: round 3's kernel swept a pair's segments back to back, so an XCD's wavefronts were in both segments of a
// 768-column profile at once -- a 4.2 MB table against a 4 MB L2, hit rate 0.77, 0.6 of the one-wavefront R = 6
// class's rate (profiles/r03/c3_rowsweep_segsweep_pmc_derived.txt).  Segment-major, the pairs an XCD works on share
// ONE segment's table (2.1 MB at R = 6, 2.8 MB at R = 8), the working set of a one-wavefront class.  The kernel
// boundary is the hand-off between segments: the column a launch reads was written by the launch before it, so there
// is no intra-kernel store -> scalar-load ordering to rely on (ADVICE r3 on round 3's form).
// Grid mode only; a persistent grid strides over the class's profiles x queries.  The LAST segment's launch has
// E(j) and J(j): pairs with E -> B / J -> B feedback, and all pairs of a profile flagged DCP_PROF_EXACT_E, are
// appended to `seg_redo` for the exact kernel (pair mode), which runs behind it on the same stream.
// ============================================================================
// Like the one-wavefront classes' large-batch kernels, a block is four wavefronts scoring consecutive queries
// against ONE profile's segment, whose 20 leading emission rows (the one- and two-base words: two of a row's five
// reads) they copy to LDS together -- the segment's 64 x R columns plus the few of the row's -inf tail a lane past
// the last node points at.  Two wavefronts per SIMD (no spills in the row loops).
#ifndef DCP_SEG_WAVES5
#define DCP_SEG_WAVES5 3 // R = 5 (profiles of 513 .. 640 nodes)
#endif
#ifndef DCP_SEG_WAVES
#define DCP_SEG_WAVES 2 // R = 6 at three (168 VGPRs, 82 spilled, 34 scratch operations per five rows of the last segment's loop): R3W4 266 ms against 256
#endif
constexpr int kSegStaged = 20;      // rows of the segment's image in LDS
constexpr unsigned kSegStagePad = 8; // columns past the segment's 64 x R that a staged row also holds
template <int R>
__global__ __launch_bounds__(256, R == 5 ? DCP_SEG_WAVES5 : DCP_SEG_WAVES) void viterbi_segment_kernel(dcp_scan_args a)
{
    constexpr unsigned SLD = 64u * R + kSegStagePad; // floats per staged row
    __shared__ __attribute__((aligned(16))) float stage_mem[kSegStaged * SLD];
    unsigned const lane = threadIdx.x & 63u;
    unsigned const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned const nblk = gridDim.x; // multiple of 8
    // XCD-aware block map as in the row sweep: an XCD's blocks take one contiguous range of tasks = consecutive
    // queries of the same few profiles, whose segment table then stays in that XCD's L2
    unsigned const vblk = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    unsigned const bpp = (a.seg_nq + 3u) / 4u; // blocks per profile: four queries each
    unsigned const seg = a.seg_index;          // this launch's segment
    float const ni = neg_inf();
    // (a do-while(false): `continue` leaves the block's one task)
    for (unsigned once = 0; once < 1u && vblk < a.nprof * bpp; ++once)
    {
        unsigned const s_rel = vblk / bpp;
        unsigned const slot = __builtin_amdgcn_readfirstlane(a.first_prof + s_rel);
        unsigned const qi = (vblk - s_rel * bpp) * 4u + wave; // this wavefront's query of the chunk
        bool const has_q = qi < a.seg_nq;                      // a spare wavefront only helps with the copy
        unsigned const q = __builtin_amdgcn_readfirstlane(a.seg_q0 + (has_q ? qi : 0u));
        unsigned const task = s_rel * a.seg_nq + qi;           // the pair's column
        dcp_prof_meta const pm = a.profs[slot];
        unsigned const ldk = pm.ldk; // core_size + 8 rounded up to 4, at most the class capacity
        unsigned const nseg = (pm.core_size + 64u * R - 1u) / (64u * R);
        if (seg >= nseg) continue; // a shorter profile of the class: done in an earlier launch (block-uniform)
        bool const last = seg + 1u == nseg;
        bool const flagged = (pm.flags & DCP_PROF_EXACT_E) != 0u;
        if (flagged)
        {
            // E(j) is not the match states' maximum for this profile: the exact kernel scores all its pairs
            if (last && has_q && lane == 0u)
            {
                unsigned const h = atomicAdd(a.seg_redo_n, 1u);
                if (h < a.seg_redo_cap) a.seg_redo[h] = dcp_pair{q, slot};
            }
            continue;
        }
        float const *__restrict__ em_prof = a.emis_match + pm.emis_off;
        unsigned const col0 = seg * 64u * R; // the segment's first column
        {
            // the segment's columns of the table's first kSegStaged rows: [code][SLD] in LDS
            unsigned const ncols = min(ldk - col0, SLD); // multiple of 4 (ldk and col0 are)
            unsigned const n4 = ncols >> 2;
            for (unsigned i = threadIdx.x; i < (unsigned)kSegStaged * n4; i += 256u)
            {
                unsigned const c = i / n4, k = i - c * n4;
                reinterpret_cast<float4 *>(stage_mem + c * SLD)[k] =
                    reinterpret_cast<float4 const *>(em_prof + (size_t)c * ldk + col0)[k];
            }
            __syncthreads();
        }
        if (!has_q) continue;
        cfloat *eN_tab = as_const(a.emis_null + (size_t)pm.pidx * DCP_NCODES);
        cfloat *eI_tab = as_const(a.emis_insert + (size_t)pm.pidx * DCP_NCODES);
        unsigned const L = a.seq_len[q];
        cu32 *words = as_const(a.seq_words + a.seq_woff[q]);
        cfloat *xt = as_const(a.xtrans + (size_t)q * DCP_XSTRIDE);
        unsigned const x = lane & 3u; // this lane's special state: 0 N, 1 J, 2 C, 3 R
        LaneSpecial sp;
        sp.a = x == 1u ? xt[DCP_X_EJ] : x == 2u ? xt[DCP_X_EC] : ni;
        sp.b = x == 0u ? xt[DCP_X_NN] : x == 1u ? xt[DCP_X_JJ] : x == 2u ? xt[DCP_X_CC] : xt[DCP_X_RR];
        sp.c = x == 0u ? xt[DCP_X_NB] : ni;          // B(j) as swept: N(j) + NB
        float const cJ = x == 1u ? xt[DCP_X_JB] : ni; // the J -> B candidate, checked in the last segment
        float xEB = xt[DCP_X_EB];
        asm volatile("" : "+v"(xEB));
        RowOut o{ni, ni};
        unsigned const node0 = col0 + lane * R;
        // a lane past the last node reads R columns of the row's -inf tail: the first aligned ones, which the staged
        // image holds too (at most kSegStagePad past the segment's columns); a row clipped to the class capacity
        // (no full tail) ends in them
        unsigned const m4 = (pm.core_size + 3u) & ~3u;
        unsigned const lane_off = node0 < pm.core_size ? node0 : (m4 + R <= ldk ? m4 : ldk - R);
        float const *__restrict__ em_base = em_prof; // wave-uniform; the lane's columns through lane_boff
        Trans<R> t;
        {
            float const *__restrict__ tb = a.trans8 + pm.trans_off + lane_off;
            VecLoad<R>::ld(tb + (size_t)DCP_T_ENTRY * ldk, t.ent);
            VecLoad<R>::ld(tb + (size_t)DCP_T_MM * ldk, t.mm);
            VecLoad<R>::ld(tb + (size_t)DCP_T_IM * ldk, t.im);
            VecLoad<R>::ld(tb + (size_t)DCP_T_DM * ldk, t.dm);
            VecLoad<R>::ld(tb + (size_t)DCP_T_MD * ldk, t.md);
            VecLoad<R>::ld(tb + (size_t)DCP_T_DD * ldk, t.dd);
            VecLoad<R>::ld(tb + (size_t)DCP_T_MI * ldk, t.mi);
            VecLoad<R>::ld(tb + (size_t)DCP_T_II * ldk, t.ii);
        }
        // The pair's boundary columns: row j's {m, i, d, e} at [j].  What segment s - 1 wrote is read with SCALAR
        // loads (wave-uniform values that only lane 0 needs as operands: they count on lgkmcnt with the row's other
        // scalar loads, so waiting for them never drains the emission rows in flight); it was written by the launch
        // before this one.  Segment s writes the pair's other column.
        size_t const col = (size_t)task * a.seg_stride;
        cfloat *bsrc = as_const((seg & 1u ? a.seg_col1 : a.seg_col0) + 4u * col);
        float4 *bdst = reinterpret_cast<float4 *>(seg & 1u ? a.seg_col0 : a.seg_col1) + col;
        PairState<R> s;
#pragma unroll
        for (int h = 0; h < 5; ++h)
        {
#pragma unroll
            for (int r = 0; r < R; ++r)
                s.P[h][r] = ni, s.Q[h][r] = ni;
            s.PX[h] = ni;
        }
        {
            float const B0 = 0.0f + xt[DCP_X_SB];
#pragma unroll
            for (int r = 0; r < R; ++r)
                s.P[0][r] = B0 + t.ent[r];
            s.PX[0] = x == 0u ? 0.0f + xt[DCP_X_SN] : x == 3u ? 0.0f : ni;
        }
        float em[5][R], eN[5], eI[5];
        unsigned w = base_at(words, 0);
        unsigned lane_boff = lane_off * 4u;
        // the staged image starts at the segment's first column: the same lane offset minus col0 (as a pointer bias)
        float const *const stg = stage_mem - col0;
        load_row<R, kSegStaged>(em_base, ldk, lane_boff, eN_tab, eI_tab, w, em, em[2], em[3], em[4], eN, eI, stg, SLD);
        SegBnd bnd{ni, ni, ni, ni}, bnx{ni, ni, ni, ni};
        if (seg > 0u) bnd = SegBnd{bsrc[4], bsrc[5], bsrc[6], bsrc[7]}; // row 1
        unsigned j = 1;
        bool seg_dirty = false;
#define DCP_SROW(PH, LASTSEG)                                                                      \
    {                                                                                             \
        w = ((w << 2) | base_at(words, j)) & 1023u;                                               \
        o = seg_row<R, PH, LASTSEG>(s, t, em, eN, eI, sp, cJ, xEB, bnd, bdst + j, lane, seg_dirty, [&]() { \
            load_row<R, kSegStaged>(em_base, ldk, lane_boff, eN_tab, eI_tab, w, em, em[2], em[3], em[4], eN, eI, stg, SLD); \
            if (seg > 0u) /* row L + 1 exists (never written: garbage, unused) */                  \
                bnx = SegBnd{bsrc[4u * j + 4u], bsrc[4u * j + 5u], bsrc[4u * j + 6u], bsrc[4u * j + 7u]}; \
        });                                                                                       \
        bnd = bnx;                                                                                \
        ++j;                                                                                      \
    }
        if (last)
        {
            while (j + 4 <= L)
            {
                DCP_SROW(1, true) DCP_SROW(2, true) DCP_SROW(3, true) DCP_SROW(4, true) DCP_SROW(0, true)
            }
            if (j <= L) DCP_SROW(1, true)
            if (j <= L) DCP_SROW(2, true)
            if (j <= L) DCP_SROW(3, true)
            if (j <= L) DCP_SROW(4, true)
        }
        else
        {
            while (j + 4 <= L)
            {
                DCP_SROW(1, false) DCP_SROW(2, false) DCP_SROW(3, false) DCP_SROW(4, false) DCP_SROW(0, false)
            }
            if (j <= L) DCP_SROW(1, false)
            if (j <= L) DCP_SROW(2, false)
            if (j <= L) DCP_SROW(3, false)
            if (j <= L) DCP_SROW(4, false)
        }
#undef DCP_SROW
        if (!last) continue;
        if (__any(seg_dirty)) // wave-uniform
        {
            if (lane == 0u)
            {
                unsigned const h = atomicAdd(a.seg_redo_n, 1u);
                if (h < a.seg_redo_cap) a.seg_redo[h] = dcp_pair{q, slot};
            }
            continue;
        }
        float const C = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o.X), 2));
        float const nul = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o.X), 3));
        float const alt = fmaxf(o.E + xt[DCP_X_ET], C + xt[DCP_X_CT]);
        if (lane == 0u)
        {
            size_t const oi = (size_t)q * a.nprof_total + pm.pidx;
            if (a.out_null) a.out_null[oi] = nul;
            if (a.out_alt) a.out_alt[oi] = alt;
            float const lrt = -2 * (nul - alt);
            if (__builtin_isfinite(lrt) && !(lrt < a.lrt_threshold))
            {
                unsigned const h = atomicAdd(a.nhits, 1u);
                if (h < a.hit_cap) a.hits[h] = dcp_hit{a.q_base + q, pm.pidx, nul, alt};
            }
        }
    }
}

// ============================================================================
// Alt-path traceback for hits.  One wavefront per hit:
//   forward: the same recursion, every M/I/D value and the specials stored in a
//            per-hit work area (float32 [L+1][ldk] x 3 + 5 x [L+1]);
//   backward: lane 0 walks from T(L) to S(0) re-deriving each arg-max from the
//            stored values -- first maximum wins, candidates in the order the
//            reference wires the transitions (src/model/protein_model.c:460-500,
//            :322-340), which is also the oracle's order.
// Produces imm_path steps {state_id, seqlen} (include/deciphon/model/protein_state.h).
// Not a throughput kernel: hits are rare (lrt >= 10).
// ============================================================================
namespace
{
struct TraceView
{
    float *Mv, *Iv, *Dv; // [L+1][ldk]
    float *N, *B, *E, *J, *C; // [L+1]
    float const *ent, *mm, *im, *dm, *md, *dd, *mi, *ii;
    float const *eM, *eI, *eN;
    float const *xt;
    uint32_t const *words;
    unsigned ldk, wd, M, L; // ldk: row length of the tables (shared rows: the group's); wd: the profile's own columns = row length of Mv / Iv / Dv
};

__device__ __forceinline__ unsigned window_at(uint32_t const *words, unsigned j)
{
    // rolling base-4 value of the (up to) 5 bases ending at position j (1-based rows)
    unsigned w = 0;
    unsigned const lo = j > 5u ? j - 5u : 0u;
    for (unsigned p = lo; p < j; ++p)
        w = (w << 2) | base_at(words, p);
    return w & 1023u;
}

// P_k(jj): best predecessor of M_k leaving row jj; *arg: 0 = M_{k-1}, 1 = I_{k-1}, 2 = D_{k-1}, 3 = B
__device__ float trace_P(TraceView const &v, unsigned jj, unsigned k, int *arg)
{
    float best = -__builtin_inff();
    int a = -1;
    if (k > 0)
    {
        size_t const o = (size_t)jj * v.wd + k - 1;
        float c0 = v.Mv[o] + v.mm[k], c1 = v.Iv[o] + v.im[k], c2 = v.Dv[o] + v.dm[k];
        if (c0 > best) best = c0, a = 0;
        if (c1 > best) best = c1, a = 1;
        if (c2 > best) best = c2, a = 2;
    }
    float const cb = v.B[jj] + v.ent[k];
    if (cb > best) best = cb, a = 3;
    if (arg) *arg = a;
    return best;
}

// Q_k(jj): best predecessor of I_k; *arg: 0 = M_k, 1 = I_k
__device__ float trace_Q(TraceView const &v, unsigned jj, unsigned k, int *arg)
{
    size_t const o = (size_t)jj * v.wd + k;
    float best = -__builtin_inff();
    int a = -1;
    float c0 = v.Mv[o] + v.mi[k], c1 = v.Iv[o] + v.ii[k];
    if (c0 > best) best = c0, a = 0;
    if (c1 > best) best = c1, a = 1;
    if (arg) *arg = a;
    return best;
}

// predecessor maxima of the special emitting states; *arg 0 = first source, 1 = self loop
__device__ float trace_PN(TraceView const &v, unsigned jj, int *arg)
{
    float best = -__builtin_inff();
    int a = -1;
    float c0 = (jj == 0 ? 0.0f : -__builtin_inff()) + v.xt[DCP_X_SN];
    float c1 = v.N[jj] + v.xt[DCP_X_NN];
    if (c0 > best) best = c0, a = 0;
    if (c1 > best) best = c1, a = 1;
    if (arg) *arg = a;
    return best;
}
__device__ float trace_PJ(TraceView const &v, unsigned jj, int *arg)
{
    float best = -__builtin_inff();
    int a = -1;
    float c0 = v.E[jj] + v.xt[DCP_X_EJ], c1 = v.J[jj] + v.xt[DCP_X_JJ];
    if (c0 > best) best = c0, a = 0;
    if (c1 > best) best = c1, a = 1;
    if (arg) *arg = a;
    return best;
}
__device__ float trace_PC(TraceView const &v, unsigned jj, int *arg)
{
    float best = -__builtin_inff();
    int a = -1;
    float c0 = v.E[jj] + v.xt[DCP_X_EC], c1 = v.C[jj] + v.xt[DCP_X_CC];
    if (c0 > best) best = c0, a = 0;
    if (c1 > best) best = c1, a = 1;
    if (arg) *arg = a;
    return best;
}
} // namespace

__global__ __launch_bounds__(64) void viterbi_trace_kernel(dcp_trace_args a)
{
    unsigned const lane = threadIdx.x;
    unsigned const h = blockIdx.x;
    if (h >= a.nhits) return;
    dcp_hit const hit = a.hits[h];
    dcp_prof_meta const pm = a.profs[a.slot_of_pidx[hit.profile_idx]];
    unsigned const q = hit.seq_idx;
    float const ni = neg_inf();

    TraceView v;
    v.ldk = pm.ldk;
    v.wd = a.work_ld ? a.work_ld[h] : pm.width; // rows of the work matrices: 64 R W of the class after the row sweep's forward pass
    v.M = pm.core_size;
    v.L = a.seq_len[q];
    v.words = a.seq_words + a.seq_woff[q];
    v.xt = a.xtrans + (size_t)q * DCP_XSTRIDE;
    float const *tb = a.trans8 + pm.trans_off;
    v.ent = tb + (size_t)DCP_T_ENTRY * v.ldk, v.mm = tb + (size_t)DCP_T_MM * v.ldk;
    v.im = tb + (size_t)DCP_T_IM * v.ldk, v.dm = tb + (size_t)DCP_T_DM * v.ldk;
    v.md = tb + (size_t)DCP_T_MD * v.ldk, v.dd = tb + (size_t)DCP_T_DD * v.ldk;
    v.mi = tb + (size_t)DCP_T_MI * v.ldk, v.ii = tb + (size_t)DCP_T_II * v.ldk;
    v.eM = a.emis_match + pm.emis_off;
    v.eI = a.emis_insert + (size_t)pm.pidx * DCP_NCODES;
    v.eN = a.emis_null + (size_t)pm.pidx * DCP_NCODES;
    unsigned const L = v.L, ldk = v.wd; // (the work arrays' row length: the profile's own columns)
    size_t const mat = (size_t)(L + 1) * ldk;
    float *work = a.work + a.work_off[h];
    v.Mv = work, v.Iv = work + mat, v.Dv = work + 2 * mat;
    v.N = work + 3 * mat, v.B = v.N + (L + 1), v.E = v.B + (L + 1), v.J = v.E + (L + 1), v.C = v.J + (L + 1);

    // ---- null model (one state R, src/model/protein_model.c:223-225,316-320) --------------
    if (a.null_model)
    {
        if (lane != 0) return;
        float *Rv = v.N; // [L+1]
        Rv[0] = ni;
        auto PR = [&](unsigned jj) { return jj == 0 ? 0.0f : Rv[jj] + v.xt[DCP_X_RR]; };
        for (unsigned j = 1; j <= L; ++j)
        {
            unsigned const w = window_at(v.words, j);
            float r = ni;
            for (unsigned l = 1; l <= (j < 5u ? j : 5u); ++l)
                r = fmaxf(r, PR(j - l) + v.eN[code_of(w, (int)l)]);
            Rv[j] = r;
        }
        a.alt_out[h] = Rv[L];
        dcp_step *out = a.steps + a.step_off[h];
        unsigned const cap = a.step_off[h + 1] - a.step_off[h];
        unsigned n = 0, j = L;
        bool ok = Rv[L] > ni;
        while (ok && j > 0)
        {
            unsigned const w = window_at(v.words, j);
            float best = ni;
            unsigned bl = 0;
            for (unsigned l = 1; l <= (j < 5u ? j : 5u); ++l)
            {
                float sc = PR(j - l) + v.eN[code_of(w, (int)l)];
                if (sc > best) best = sc, bl = l;
            }
            if (bl == 0) { ok = false; break; }
            if (n < cap) out[n] = dcp_step{(uint16_t)((3u << 14) | 0u), (uint8_t)bl, 0};
            ++n;
            j -= bl;
        }
        unsigned const m = n < cap ? n : cap;
        for (unsigned i = 0; i < m / 2; ++i)
        {
            dcp_step t = out[i];
            out[i] = out[m - 1 - i];
            out[m - 1 - i] = t;
        }
        a.nsteps[h] = ok ? n : 0xffffffffu;
        return;
    }

    // ---- forward -----------------------------------------------------------------
    // (skipped when viterbi_rowsweep_kernel<..., TRACE> has filled the work area: dcp_gpu_trace_paths' default; this
    // one-wavefront loop over rows in global memory is the tests' second implementation of it)
    if (!a.skip_forward)
    {
    // nodes per lane, contiguous; a one-wavefront profile's rows are not a multiple of 64 columns wide
    // (dcp_gpu_db_upload), so the last lanes may own fewer nodes or none
    unsigned const R = (ldk + 63u) / 64u;
    unsigned const k0 = lane * R;
    unsigned const nr = k0 >= ldk ? 0u : (ldk - k0 < R ? ldk - k0 : R); // this lane's nodes
    for (unsigned r = 0; r < nr; ++r)
        v.Mv[k0 + r] = v.Iv[k0 + r] = v.Dv[k0 + r] = ni; // row 0
    if (lane == 0)
    {
        v.N[0] = ni, v.E[0] = ni, v.J[0] = ni, v.C[0] = ni;
        v.B[0] = 0.0f + v.xt[DCP_X_SB];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();

    for (unsigned j = 1; j <= L; ++j)
    {
        unsigned const w = window_at(v.words, j);
        unsigned const maxl = j < 5u ? j : 5u;
        float *Mj = v.Mv + (size_t)j * ldk, *Ij = v.Iv + (size_t)j * ldk, *Dj = v.Dv + (size_t)j * ldk;
        for (unsigned r = 0; r < nr; ++r)
        {
            unsigned const k = k0 + r;
            float m = ni, iv = ni;
            for (unsigned l = 1; l <= maxl; ++l)
            {
                unsigned const c = code_of(w, (int)l);
                m = fmaxf(m, trace_P(v, j - l, k, nullptr) + v.eM[(size_t)c * v.ldk + k]);
                iv = fmaxf(iv, trace_Q(v, j - l, k, nullptr) + v.eI[c]);
            }
            Mj[k] = m, Ij[k] = iv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        // delete chain to the fixed point (same construction as the scoring kernels)
        float d_last = ni;
        for (;;)
        {
            float const before = d_last;
            float d = lane_shr1(d_last, ni); // D of node k0-1
            for (unsigned r = 0; r < nr; ++r)
            {
                unsigned const k = k0 + r;
                d = k == 0 ? ni : fmaxf(Mj[k - 1] + v.md[k], d + v.dd[k]);
                Dj[k] = d;
            }
            d_last = nr ? d : ni; // a lane without nodes passes nothing on
            if (!__any(d_last != before)) break;
        }
        float e = ni;
        for (unsigned r = 0; r < nr; ++r)
            e = fmaxf(e, fmaxf(Mj[k0 + r], Dj[k0 + r]));
        float const E = wave_max(e);
        float N = ni, J = ni, C = ni;
        for (unsigned l = 1; l <= maxl; ++l)
        {
            float const en = v.eN[code_of(w, (int)l)];
            N = fmaxf(N, trace_PN(v, j - l, nullptr) + en);
            J = fmaxf(J, trace_PJ(v, j - l, nullptr) + en);
            C = fmaxf(C, trace_PC(v, j - l, nullptr) + en);
        }
        if (lane == 0)
        {
            v.N[j] = N, v.E[j] = E, v.J[j] = J, v.C[j] = C;
            v.B[j] = fmaxf(fmaxf(N + v.xt[DCP_X_NB], E + v.xt[DCP_X_EB]), J + v.xt[DCP_X_JB]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
    }
    }

    // ---- backward (lane 0) ---------------------------------------------------------
    if (lane != 0) return;
    float const alt = fmaxf(v.E[L] + v.xt[DCP_X_ET], v.C[L] + v.xt[DCP_X_CT]);
    a.alt_out[h] = alt;
    dcp_step *out = a.steps + a.step_off[h];
    unsigned const cap = a.step_off[h + 1] - a.step_off[h];
    unsigned n = 0;
    bool ok = alt > ni;
    enum { ST_S = 1, ST_N, ST_B, ST_E, ST_J, ST_C, ST_T, ST_M, ST_I, ST_D };
    int st = ST_T;
    unsigned k = 0, j = L;
    auto push = [&](unsigned id, unsigned len) {
        if (n < cap) out[n] = dcp_step{(uint16_t)id, (uint8_t)len, 0};
        ++n;
    };
    unsigned const EXT = 3u << 14;
    unsigned guard = 0, const_guard = 4u * (L + v.M) + 64u;
    while (ok && guard++ < const_guard)
    {
        if (st == ST_T)
        {
            push(EXT | 7u, 0);
            float c0 = v.E[j] + v.xt[DCP_X_ET], c1 = v.C[j] + v.xt[DCP_X_CT];
            st = !(c1 > c0) ? ST_E : ST_C; // E->T was wired first
        }
        else if (st == ST_E)
        {
            push(EXT | 4u, 0);
            // M_M, M_1..M_{M-1}, D_2..D_M (1-based): protein_model.c:494, :441-458
            float best = ni;
            int bs = -1;
            unsigned bk = 0;
            float const *Mj = v.Mv + (size_t)j * ldk, *Dj = v.Dv + (size_t)j * ldk;
            float c = Mj[v.M - 1] + 0.0f;
            if (c > best) best = c, bs = ST_M, bk = v.M - 1;
            for (unsigned kk = 0; kk + 1 < v.M; ++kk)
            {
                c = Mj[kk] + 0.0f;
                if (c > best) best = c, bs = ST_M, bk = kk;
            }
            for (unsigned kk = 1; kk < v.M; ++kk)
            {
                c = Dj[kk] + 0.0f;
                if (c > best) best = c, bs = ST_D, bk = kk;
            }
            if (bs < 0) ok = false;
            st = bs, k = bk;
        }
        else if (st == ST_M || st == ST_I || st == ST_N || st == ST_J || st == ST_C)
        {
            unsigned const w = window_at(v.words, j);
            unsigned const maxl = j < 5u ? j : 5u;
            float best = ni;
            unsigned bl = 0;
            for (unsigned l = 1; l <= maxl; ++l)
            {
                unsigned const c = code_of(w, (int)l);
                float sc;
                if (st == ST_M) sc = trace_P(v, j - l, k, nullptr) + v.eM[(size_t)c * v.ldk + k];
                else if (st == ST_I) sc = trace_Q(v, j - l, k, nullptr) + v.eI[c];
                else if (st == ST_N) sc = trace_PN(v, j - l, nullptr) + v.eN[c];
                else if (st == ST_J) sc = trace_PJ(v, j - l, nullptr) + v.eN[c];
                else sc = trace_PC(v, j - l, nullptr) + v.eN[c];
                if (sc > best) best = sc, bl = l;
            }
            if (bl == 0) { ok = false; break; }
            int arg = -1;
            if (st == ST_M)
            {
                push(k + 1u, bl);
                trace_P(v, j - bl, k, &arg);
                j -= bl;
                if (arg == 0) st = ST_M, k = k - 1;
                else if (arg == 1) st = ST_I, k = k - 1;
                else if (arg == 2) st = ST_D, k = k - 1;
                else if (arg == 3) st = ST_B;
                else ok = false;
            }
            else if (st == ST_I)
            {
                push((1u << 14) | (k + 1u), bl);
                trace_Q(v, j - bl, k, &arg);
                j -= bl;
                if (arg == 0) st = ST_M;
                else if (arg == 1) st = ST_I;
                else ok = false;
            }
            else if (st == ST_N)
            {
                push(EXT | 2u, bl);
                trace_PN(v, j - bl, &arg);
                j -= bl;
                st = arg == 0 ? ST_S : ST_N;
                if (arg < 0) ok = false;
            }
            else if (st == ST_J)
            {
                push(EXT | 5u, bl);
                trace_PJ(v, j - bl, &arg);
                j -= bl;
                st = arg == 0 ? ST_E : ST_J;
                if (arg < 0) ok = false;
            }
            else
            {
                push(EXT | 6u, bl);
                trace_PC(v, j - bl, &arg);
                j -= bl;
                st = arg == 0 ? ST_E : ST_C;
                if (arg < 0) ok = false;
            }
        }
        else if (st == ST_D)
        {
            push((2u << 14) | (k + 1u), 0);
            if (k == 0) { ok = false; break; }
            size_t const o = (size_t)j * ldk + k - 1;
            float c0 = v.Mv[o] + v.md[k], c1 = v.Dv[o] + v.dd[k];
            st = !(c1 > c0) ? ST_M : ST_D; // M_{k-1} -> D_k was wired first
            k = k - 1;
        }
        else if (st == ST_B)
        {
            push(EXT | 3u, 0);
            // S->B, N->B, E->B, J->B (protein_model.c:324-337)
            float best = ni;
            int bs = -1;
            float c = (j == 0 ? 0.0f : ni) + v.xt[DCP_X_SB];
            if (c > best) best = c, bs = ST_S;
            c = v.N[j] + v.xt[DCP_X_NB];
            if (c > best) best = c, bs = ST_N;
            c = v.E[j] + v.xt[DCP_X_EB];
            if (c > best) best = c, bs = ST_E;
            c = v.J[j] + v.xt[DCP_X_JB];
            if (c > best) best = c, bs = ST_J;
            if (bs < 0) ok = false;
            st = bs;
        }
        else // ST_S
        {
            push(EXT | 1u, 0);
            if (j != 0) ok = false;
            break;
        }
    }
    if (st != ST_S) ok = false;
    unsigned const m = n < cap ? n : cap;
    for (unsigned i = 0; i < m / 2; ++i)
    {
        dcp_step t = out[i];
        out[i] = out[m - 1 - i];
        out[m - 1 - i] = t;
    }
    a.nsteps[h] = ok ? n : 0xffffffffu;
}

extern "C" void dcp_launch_trace(dcp_trace_args const *a, unsigned nhits, void *stream)
{
    hipLaunchKernelGGL(viterbi_trace_kernel, dim3(nhits), dim3(64), 0, (hipStream_t)stream, *a);
}

// ============================================================================
// Emission-table expansion: out[code][k] for a tile of 64 nodes per block.
// Same probability-domain formula as dcp_frame_table_host (dcp_model.cpp).
// ============================================================================
namespace
{
__device__ __forceinline__ double c3(double const *C, int x, int y, int z)
{
    return C[x * 25 + y * 5 + z];
}
__device__ __forceinline__ double s1(double const *C, int x)
{
    return c3(C, x, 4, 4) + c3(C, 4, x, 4) + c3(C, 4, 4, x);
}
__device__ __forceinline__ double s2(double const *C, int x, int y)
{
    return c3(C, 4, x, y) + c3(C, x, 4, y) + c3(C, x, y, 4);
}

__device__ double frame_prob(double const *b, double const *C, double e, double f,
                             unsigned code)
{
    double const e2 = e * e, f2 = f * f;
    if (code < 4) return e2 * f2 / 3.0 * s1(C, (int)code);
    if (code < 20)
    {
        int v = (int)code - 4, x1 = v >> 2, x2 = v & 3;
        return 2.0 * e * f2 * f / 3.0 * s2(C, x1, x2) +
               e2 * e * f / 3.0 * (b[x2] * s1(C, x1) + b[x1] * s1(C, x2));
    }
    if (code < 84)
    {
        int v = (int)code - 20, x1 = v >> 4, x2 = (v >> 2) & 3, x3 = v & 3;
        return f2 * f2 * c3(C, x1, x2, x3) +
               4.0 * e2 * f2 / 9.0 *
                   (b[x1] * s2(C, x2, x3) + b[x2] * s2(C, x1, x3) +
                    b[x3] * s2(C, x1, x2)) +
               e2 * e2 / 9.0 *
                   (b[x1] * b[x2] * s1(C, x3) + b[x1] * b[x3] * s1(C, x2) +
                    b[x2] * b[x3] * s1(C, x1));
    }
    if (code < 340)
    {
        int v = (int)code - 84;
        int x[4] = {v >> 6, (v >> 4) & 3, (v >> 2) & 3, v & 3};
        double one = b[x[0]] * c3(C, x[1], x[2], x[3]) + b[x[1]] * c3(C, x[0], x[2], x[3]) +
                     b[x[2]] * c3(C, x[0], x[1], x[3]) + b[x[3]] * c3(C, x[0], x[1], x[2]);
        double two = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j)
            {
                int r[2], n = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k != i && k != j) r[n++] = x[k];
                two += b[x[i]] * b[x[j]] * s2(C, r[0], r[1]);
            }
        return e * f2 * f / 2.0 * one + e2 * e * f / 9.0 * two;
    }
    {
        int v = (int)code - 340;
        int x[5] = {v >> 8, (v >> 6) & 3, (v >> 4) & 3, (v >> 2) & 3, v & 3};
        double s = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = i + 1; j < 5; ++j)
            {
                int r[3], n = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if (k != i && k != j) r[n++] = x[k];
                s += b[x[i]] * b[x[j]] * c3(C, r[0], r[1], r[2]);
            }
        return e2 * f2 / 10.0 * s;
    }
}
} // namespace

// grid.x = number of 64-node tiles over all tables; tiles[] says where each
// one reads its dists and writes its columns.
__global__ __launch_bounds__(256) void expand_tables_kernel(dcp_expand_args a)
{
    __shared__ double lin[64][DCP_NDIST + 1]; // +1: odd stride in 8-byte words
    dcp_expand_tile const tile = a.tiles[blockIdx.x];
    float const *__restrict__ dist = a.dists + (size_t)tile.dist_row * DCP_NDIST;
    unsigned const nk = tile.ncols; // 0..64 columns backed by a dist row
    for (unsigned i = threadIdx.x; i < 64u * DCP_NDIST; i += 256u)
    {
        unsigned k = i / DCP_NDIST, c = i - k * DCP_NDIST;
        lin[k][c] = k < nk ? exp((double)dist[(size_t)k * DCP_NDIST + c]) : 0.0;
    }
    __syncthreads();
    unsigned const k = threadIdx.x & 63u;
    double const e = k < nk ? (double)a.eps[tile.dist_row + k] : 0.0, f = 1.0 - e;
    size_t col_off = (size_t)k * tile.ld_col;
    unsigned ld_code = tile.ld_code;
    if (tile.kt)
    {
        unsigned const node = tile.col0 + k, kt = tile.kt;
        col_off = ((size_t)(node / kt) * (kt / 4u) + (node % kt) / 4u) * (size_t)DCP_NCODES * 4u + (node & 3u);
        ld_code = 4u;
    }
    float *__restrict__ out = a.out + tile.out_off + col_off;
    for (unsigned code = threadIdx.x >> 6; code < DCP_NCODES; code += 4u)
    {
        float v = -__builtin_inff(); // padding columns: unreachable nodes
        if (k < nk) v = (float)log(frame_prob(&lin[k][0], &lin[k][4], e, f, code));
        if (k < tile.nstore) out[(size_t)code * ld_code] = v;
    }
}

// ---- launchers ------------------------------------------------------------
extern "C" void dcp_launch_expand(dcp_expand_args const *a, unsigned ntiles,
                                  void *stream)
{
    hipLaunchKernelGGL(expand_tables_kernel, dim3(ntiles), dim3(256), 0,
                       (hipStream_t)stream, *a);
}

template <int R, int W, int STG, bool PF = false>
static void launch_rs(dcp_scan_args const *a, unsigned nblocks, unsigned threads, hipStream_t s, unsigned pad_lds = 0)
{
    // pad_lds: unused dynamic LDS -- fewer blocks per CU (an occupancy experiment of the tests' build)
    hipLaunchKernelGGL((viterbi_rowsweep_kernel<R, W, STG, PF>), dim3(nblocks), dim3(threads), pad_lds, s, *a);
}

// tasks per block of the unstaged (R, W) kernel: 4 independent wavefronts when W == 1
extern "C" unsigned dcp_rowsweep_tasks_per_block(int W) { return W == 1 ? 4u : 1u; }

// Is there a kernel that stages `stg` rows for this class, and how many wavefronts may its block have?
extern "C" unsigned dcp_rowsweep_max_block_waves(int R, int W, int stg)
{
    if (W != 1) return stg == 0 ? 1u : 0u;
    if (stg == 0) return 4u;
    // (the narrower of the variant with and without the two-row prefetch: one answer per class and image)
    if (stg == 20) return R >= 1 && R <= 8 ? (unsigned)std::min(rs_block_threads(R, 1, 20, false), rs_block_threads(R, 1, 20, true)) / 64u : 0u;
    if (stg == 84) return R >= 1 && R <= 7 ? (unsigned)rs_block_threads(R, 1, 84) / 64u : 0u; // R = 8: 172 KB
    return 0u;
}
extern "C" unsigned dcp_rowsweep_stage_bytes(int R, int stg) { return (unsigned)stg * (64u * (unsigned)R + (R <= 2 ? 8u : 0u)) * 4u; }

// Grid mode: all chunks x the profiles [first_prof, first_prof + nprof) of one size class.
//   stg  rows staged in LDS (0: every wavefront its own task, four per block);
//   bw   wavefronts per staged block (1..dcp_rowsweep_max_block_waves).
extern "C" int dcp_launch_rowsweep_grid(int R, int W, dcp_scan_args const *a, int stg, unsigned bw, void *stream,
                                        unsigned pad_lds, int prefetch2)
{
    hipStream_t s = (hipStream_t)stream;
    unsigned const maxw = dcp_rowsweep_max_block_waves(R, W, stg);
    if (maxw == 0u || (stg > 0 && (bw == 0u || bw > maxw))) return -1;
    uint64_t nb;
    if (stg > 0) nb = (uint64_t)a->nprof * ((a->nchunks + bw - 1u) / bw);
    else nb = ((uint64_t)a->nprof * a->nchunks + dcp_rowsweep_tasks_per_block(W) - 1u) / dcp_rowsweep_tasks_per_block(W);
    nb = (nb + 7u) / 8u * 8u;
    if (nb > 0x7fffffffull) return -2;
    unsigned const nblocks = (unsigned)nb;
    {
        // a block's LDS ends at 160 KiB: a dispatch asking for more aborts the queue
        unsigned const used = dcp_rowsweep_stage_bytes(R, stg) + 1024u;
        pad_lds = used >= 160u * 1024u ? 0u : std::min(pad_lds, 160u * 1024u - used);
    }
#define DCP_CASE_S(r, g)                                                       \
    if (R == r && W == 1 && stg == g)                                          \
    {                                                                          \
        if (prefetch2) launch_rs<r, 1, g, true>(a, nblocks, 64u * bw, s, pad_lds); \
        else launch_rs<r, 1, g, false>(a, nblocks, 64u * bw, s, pad_lds);      \
        return 0;                                                              \
    }
    DCP_CASE_S(1, 20) DCP_CASE_S(2, 20) DCP_CASE_S(3, 20) DCP_CASE_S(4, 20) DCP_CASE_S(5, 20) DCP_CASE_S(6, 20)
    DCP_CASE_S(7, 20) DCP_CASE_S(8, 20)
    DCP_CASE_S(1, 84) DCP_CASE_S(2, 84) DCP_CASE_S(3, 84) DCP_CASE_S(4, 84) DCP_CASE_S(5, 84) DCP_CASE_S(6, 84)
    DCP_CASE_S(7, 84)
#undef DCP_CASE_S
    if (stg != 0) return -1;
    return dcp_launch_rowsweep(R, W, a, nblocks, stream);
}

extern "C" int dcp_launch_mp(int K, dcp_scan_args const *a, void *stream)
{
    uint64_t nb = (uint64_t)a->nprof * ((a->nseqs + 3u) / 4u);
    nb = (nb + 7u) / 8u * 8u;
    if (nb > 0x7fffffffull) return -2;
    if (K == 4) hipLaunchKernelGGL((viterbi_mp_kernel<4>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, *a);
    else if (K == 2) hipLaunchKernelGGL((viterbi_mp_kernel<2>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, *a);
    else return -1;
    return 0;
}
extern "C" void dcp_launch_mp_in(dcp_mp_group const *groups, unsigned ngroups, unsigned n_first, unsigned k_first, unsigned k_rest,
                                 float const *emis_insert, float const *emis_null, float *out, void *stream)
{
    hipLaunchKernelGGL(mp_in_kernel, dim3(ngroups), dim3(256), 0, (hipStream_t)stream, groups, n_first, k_first, k_rest,
                       emis_insert, emis_null, out);
}

// blocks of a segment launch: four queries of one profile each, rounded up to the XCD count
extern "C" unsigned dcp_segsweep_blocks(unsigned nprof, unsigned nq) { return (nprof * ((nq + 3u) / 4u) + 7u) / 8u * 8u; }
// One launch = segment a->seg_index of every (profile, query) pair of a->first_prof .. + a->nprof, profiles cut into
// segments of 64 x R nodes (viterbi_segment_kernel<R>, R = 5..8); != 0 if there is no such kernel.
// a->seg_col0 / seg_col1: the pairs' boundary columns, a->seg_stride rows of 16 bytes each.
extern "C" int dcp_launch_segsweep(int R, dcp_scan_args const *a, unsigned nblocks, void *stream)
{
#define DCP_CASE_G(r)                                                                                          \
    if (R == r)                                                                                                \
    {                                                                                                          \
        hipLaunchKernelGGL((viterbi_segment_kernel<r>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, *a); \
        return 0;                                                                                              \
    }
    DCP_CASE_G(5) DCP_CASE_G(6) DCP_CASE_G(7) DCP_CASE_G(8)
#undef DCP_CASE_G
    return -1;
}

// the traceback's forward pass over a->pairs: rows to a->trace_work (viterbi_rowsweep_kernel<R, W, 0, false, true>)
extern "C" int dcp_launch_trace_forward(int R, int W, dcp_scan_args const *a, unsigned nblocks, void *stream)
{
#define DCP_CASE_T(r, w)                                                                                        \
    if (R == r && W == w)                                                                                       \
    {                                                                                                           \
        hipLaunchKernelGGL((viterbi_rowsweep_kernel<r, w, 0, false, true>), dim3(nblocks), dim3(w == 1 ? 256u : 64u * w), 0, \
                           (hipStream_t)stream, *a);                                                            \
        return 0;                                                                                               \
    }
    DCP_CASE_T(1, 1) DCP_CASE_T(2, 1) DCP_CASE_T(3, 1) DCP_CASE_T(4, 1) DCP_CASE_T(5, 1) DCP_CASE_T(6, 1) DCP_CASE_T(7, 1)
    DCP_CASE_T(8, 1) DCP_CASE_T(3, 4) DCP_CASE_T(4, 4) DCP_CASE_T(3, 8) DCP_CASE_T(4, 8) DCP_CASE_T(3, 16) DCP_CASE_T(4, 16)
#undef DCP_CASE_T
    return -1;
}

// unstaged kernels: pair mode (a->pairs), the classes of several wavefronts per pair, and grid mode without staging
extern "C" int dcp_launch_rowsweep(int R, int W, dcp_scan_args const *a,
                                   unsigned nblocks, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
#define DCP_CASE(r, w)                                                         \
    if (R == r && W == w)                                                      \
    {                                                                          \
        launch_rs<r, w, 0>(a, nblocks, w == 1 ? 256u : 64u * w, s);            \
        return 0;                                                              \
    }
    DCP_CASE(1, 1) DCP_CASE(2, 1) DCP_CASE(3, 1) DCP_CASE(4, 1) DCP_CASE(5, 1) DCP_CASE(6, 1) DCP_CASE(7, 1)
    DCP_CASE(8, 1) DCP_CASE(3, 4) DCP_CASE(4, 4) DCP_CASE(3, 8) DCP_CASE(4, 8) DCP_CASE(3, 16) DCP_CASE(4, 16)
#undef DCP_CASE
    return -1;
}
