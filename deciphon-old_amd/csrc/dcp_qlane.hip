// dcp_qlane.hip -- throughput kernel of the scan path: one LANE per query.
//
// viterbi_qlane_kernel<G>: a block is 256 queries against ONE profile.  The
// profile is cut into tiles of KT = 4*G consecutive nodes; a tile's match
// emission tables (KT x 1364 floats) sit in LDS and every lane gathers the five
// rows its own sequence window selects.  A tile is swept over all rows j with
// the 5-row history of its KT nodes in registers; what the next tile needs from
// row j -- D of its first node, the M/I/D part of that node's predecessor
// maximum, the running E -- goes through per-block scratch planes in HBM
// (coalesced, 28 B per row and tile).  No cross-lane operation exists: the
// delete chain is sequential in k inside the lane.
//
// Multi-hit feedback (B(j) needs E(j) of the same row, which needs every tile):
// iteration 0 runs with B0(j) = N(j) + NB only; the last tile then knows E(j),
// J(j) and checks whether max(E(j)+EB, J(j)+JB) exceeds the B(j) that was used.
// If it never does, B0 IS the solution of the recurrence (the system is a forward
// recurrence in j, so its solution is unique) and the scores are exact; otherwise
// the improved B is stored and the pair is swept again, until nothing changes.
// Values only grow and are bounded by the true ones, so the loop ends at the
// exact Viterbi scores -- bit-identical to the row sweep of dcp_kernels.hip and
// to the oracle.  Reference recursion: imm_dp_viterbi as driven by
// src/server/scan_thread.c:99-123; model wiring src/model/protein_model.c:410-500.
#include "dcp_kernels.h"

#include <hip/hip_runtime.h>

namespace
{
constexpr int NC = DCP_NCODES;

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
    return fmaxf(fmaxf(fmaxf(a, b), fmaxf(c, d)), e);
}

__device__ __forceinline__ unsigned code_of(unsigned w, int l)
{
    constexpr unsigned off[5] = {0u, 4u, 20u, 84u, 340u};
    return off[l - 1] + (w & ((1u << (2 * l)) - 1u));
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

// One row of one tile for this lane's query.  PH = j % 5 (compile time).
template <int G, bool FIRST, bool LAST, int PH>
__device__ __forceinline__ void ql_row(QState<G> &s, cfloat *tt,
                                       float const *tabM, float const *tabI, float const *tabN,
                                       unsigned w, float *__restrict__ pB, float *__restrict__ pXm,
                                       float *__restrict__ pXd, float *__restrict__ pEm,
                                       LaneXt const &xt, bool first_iter, bool &dirty, SweepOut &o)
{
    constexpr int KT = 4 * G;
    constexpr int s1 = (PH + 4) % 5, s2 = (PH + 3) % 5, s3 = (PH + 2) % 5, s4 = (PH + 1) % 5, s5 = PH;
    float const ni = ninf();

    unsigned c[5];
#pragma unroll
    for (int l = 0; l < 5; ++l)
        c[l] = code_of(w, l + 1);

    // boundary of the previous tile and the B this iteration runs with
    float Xm = ni, Xd = ni, E = ni, Bj;
    if constexpr (!FIRST)
    {
        Xm = *pXm;
        Xd = *pXd;
        E = *pEm;
    }
    if (!FIRST || !first_iter) Bj = *pB;

    float eI[5], eN[5];
#pragma unroll
    for (int l = 0; l < 5; ++l)
        eI[l] = tabI[c[l]];
    if constexpr (FIRST || LAST)
    {
#pragma unroll
        for (int l = 0; l < 5; ++l)
            eN[l] = tabN[c[l]];
    }

    if constexpr (FIRST)
    {
        if (first_iter)
        {
            // N(j), R(j); B0(j) = N(j) + NB  (S(j>0) = -inf)
            float const N = mx5(s.PN[s1] + eN[0], s.PN[s2] + eN[1], s.PN[s3] + eN[2],
                                s.PN[s4] + eN[3], s.PN[s5] + eN[4]);
            float const Rn = mx5(s.PR[s1] + eN[0], s.PR[s2] + eN[1], s.PR[s3] + eN[2],
                                 s.PR[s4] + eN[3], s.PR[s5] + eN[4]);
            s.PN[PH] = N + xt.NN;
            s.PR[PH] = Rn + xt.RR;
            o.Rn = Rn;
            Bj = N + xt.NB;
            *pB = Bj;
        }
    }

    float pm = ni, pi = ni, pd = ni; // node k-1 of this row
#pragma unroll
    for (int g = 0; g < G; ++g)
    {
        float4 e[5];
#pragma unroll
        for (int l = 0; l < 5; ++l)
            e[l] = *reinterpret_cast<float4 const *>(tabM + ((size_t)g * NC + c[l]) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            constexpr int dummy = 0;
            (void)dummy;
            int const k = 4 * g + r;
            cfloat *tk = tt + k * 8;
            float const e0 = r == 0 ? e[0].x : r == 1 ? e[0].y : r == 2 ? e[0].z : e[0].w;
            float const e1 = r == 0 ? e[1].x : r == 1 ? e[1].y : r == 2 ? e[1].z : e[1].w;
            float const e2 = r == 0 ? e[2].x : r == 1 ? e[2].y : r == 2 ? e[2].z : e[2].w;
            float const e3 = r == 0 ? e[3].x : r == 1 ? e[3].y : r == 2 ? e[3].z : e[3].w;
            float const e4 = r == 0 ? e[4].x : r == 1 ? e[4].y : r == 2 ? e[4].z : e[4].w;
            float const m = mx5(s.P[s1][k] + e0, s.P[s2][k] + e1, s.P[s3][k] + e2, s.P[s4][k] + e3,
                                s.P[s5][k] + e4);
            float const in = mx5(s.Q[s1][k] + eI[0], s.Q[s2][k] + eI[1], s.Q[s3][k] + eI[2],
                                 s.Q[s4][k] + eI[3], s.Q[s5][k] + eI[4]);
            float d, pin;
            if (k == 0)
            {
                d = Xd;
                pin = Xm;
            }
            else
            {
                d = fmaxf(pm + tk[DCP_T_MD], pd + tk[DCP_T_DD]);
                pin = mx3(pm + tk[DCP_T_MM], pi + tk[DCP_T_IM], pd + tk[DCP_T_DM]);
            }
            E = mx3(E, m, d);
            s.P[PH][k] = fmaxf(Bj + tk[DCP_T_ENTRY], pin);
            s.Q[PH][k] = fmaxf(m + tk[DCP_T_MI], in + tk[DCP_T_II]);
            pm = m, pi = in, pd = d;
        }
    }

    if constexpr (!LAST)
    {
        cfloat *tn = tt + KT * 8; // edges into the next tile's first node
        *pXm = mx3(pm + tn[DCP_T_MM], pi + tn[DCP_T_IM], pd + tn[DCP_T_DM]);
        *pXd = fmaxf(pm + tn[DCP_T_MD], pd + tn[DCP_T_DD]);
        *pEm = E;
    }
    else
    {
        float const J = mx5(s.PJ[s1] + eN[0], s.PJ[s2] + eN[1], s.PJ[s3] + eN[2], s.PJ[s4] + eN[3],
                            s.PJ[s5] + eN[4]);
        float const C = mx5(s.PC[s1] + eN[0], s.PC[s2] + eN[1], s.PC[s3] + eN[2], s.PC[s4] + eN[3],
                            s.PC[s5] + eN[4]);
        // did E(j) -> B(j) or J(j) -> B(j) beat the B(j) this sweep ran with?
        float const B1 = fmaxf(E + xt.EB, J + xt.JB);
        if (B1 > Bj)
        {
            dirty = true;
            *pB = B1;
        }
        s.PJ[PH] = fmaxf(E + xt.EJ, J + xt.JJ);
        s.PC[PH] = fmaxf(E + xt.EC, C + xt.CC);
        o.E = E;
        o.C = C;
    }
}

// Sweep one tile over rows 1..L of this lane's query.
template <int G, bool FIRST, bool LAST>
__device__ __forceinline__ void ql_sweep(cfloat *tt, float const *tabM,
                                         float const *tabI, float const *tabN,
                                         uint32_t const *__restrict__ words, unsigned L,
                                         unsigned Lwave, bool active, float *__restrict__ sc,
                                         size_t plane, LaneXt const &xt, bool first_iter,
                                         bool &dirty, SweepOut &o)
{
    constexpr int KT = 4 * G;
    float const ni = ninf();
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
            s.P[0][k] = B0 + tt[k * 8 + DCP_T_ENTRY];
        s.PN[0] = 0.0f + xt.SN;
        s.PR[0] = 0.0f;
    }
    float *pB = sc, *pXm = sc + plane, *pXd = sc + 2 * plane, *pEm = sc + 3 * plane;
    unsigned w = 0, cur = 0, j = 1;

#define QL_ROW(PH)                                                                         \
    {                                                                                      \
        if (active && j <= L)                                                              \
        {                                                                                  \
            unsigned const pos = j - 1u;                                                   \
            if ((pos & 15u) == 0u) cur = words[pos >> 4];                                  \
            w = ((w << 2) | ((cur >> ((pos & 15u) * 2u)) & 3u)) & 1023u;                   \
            ql_row<G, FIRST, LAST, PH>(s, tt, tabM, tabI, tabN, w, pB, pXm, pXd, pEm, xt,  \
                                       first_iter, dirty, o);                              \
        }                                                                                  \
        pB += 256, pXm += 256, pXd += 256, pEm += 256;                                     \
        ++j;                                                                               \
    }
    while (j + 4 <= Lwave)
    {
        QL_ROW(1) QL_ROW(2) QL_ROW(3) QL_ROW(4) QL_ROW(0)
    }
    if (j <= Lwave) QL_ROW(1)
    if (j <= Lwave) QL_ROW(2)
    if (j <= Lwave) QL_ROW(3)
    if (j <= Lwave) QL_ROW(4)
#undef QL_ROW
}

__device__ __forceinline__ unsigned wave_umax(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        unsigned other = (unsigned)__shfl_xor((int)v, o, 64);
        v = other > v ? other : v;
    }
    return v;
}

} // namespace

template <int G>
__global__ __launch_bounds__(256, 2) void viterbi_qlane_kernel(dcp_qlane_args a)
{
    constexpr int KT = 4 * G;
    constexpr int TAB_FLOATS = G * NC * 4;
    __shared__ __attribute__((aligned(16))) float lds[TAB_FLOATS + 2 * NC];
    __shared__ unsigned s_task;
    float *tabM = lds, *tabI = lds + TAB_FLOATS, *tabN = tabI + NC;
    unsigned const tid = threadIdx.x;
    size_t const plane = (size_t)a.lmax * 256u;
    float *const sc = a.scratch + (size_t)blockIdx.x * 4u * plane + tid;

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

        unsigned const qi = qb * 256u + tid;
        bool const has = qi < a.nseqs;
        unsigned const q = has ? a.qorder[qi] : 0u;
        unsigned const L = has ? a.seq_len[q] : 0u;
        uint32_t const *__restrict__ words = a.seq_words + a.seq_woff[q];
        LaneXt xt;
        {
            float const *__restrict__ x = a.xtrans + (size_t)q * DCP_XSTRIDE;
            xt.RR = x[DCP_X_RR], xt.SB = x[DCP_X_SB], xt.SN = x[DCP_X_SN], xt.NN = x[DCP_X_NN];
            xt.NB = x[DCP_X_NB], xt.ET = x[DCP_X_ET], xt.EC = x[DCP_X_EC], xt.CC = x[DCP_X_CC];
            xt.CT = x[DCP_X_CT], xt.EB = x[DCP_X_EB], xt.EJ = x[DCP_X_EJ], xt.JJ = x[DCP_X_JJ];
            xt.JB = x[DCP_X_JB];
        }

        // the profile's insert and background tables stay in LDS for the whole task
        {
            float const *__restrict__ gi = a.emis_insert + (size_t)pm.pidx * NC;
            float const *__restrict__ gn = a.emis_null + (size_t)pm.pidx * NC;
            for (unsigned i = tid; i < (unsigned)NC; i += 256u)
                tabI[i] = gi[i], tabN[i] = gn[i];
        }

        SweepOut o{ninf(), ninf(), ninf()};
        bool dirty = false;
        bool first_iter = true;
        for (;;)
        {
            bool const active = has && (first_iter || dirty);
            unsigned const Lwave = __builtin_amdgcn_readfirstlane(wave_umax(active ? L : 0u));
            dirty = false;
            for (unsigned t = 0; t < T; ++t)
            {
                __syncthreads(); // previous tile's readers are done with tabM
                {
                    float4 const *__restrict__ src = reinterpret_cast<float4 const *>(
                        a.emis_tiles + pm.tile_off + (size_t)t * TAB_FLOATS);
                    float4 *dst = reinterpret_cast<float4 *>(tabM);
                    for (unsigned i = tid; i < (unsigned)(TAB_FLOATS / 4); i += 256u)
                        dst[i] = src[i];
                }
                __syncthreads();
                if (Lwave == 0u) continue; // no lane of this wavefront has work
                cfloat *tt = as_const(a.ttrans + pm.ttrans_off + (size_t)t * (KT + 1) * 8);
                bool const first = t == 0, last = t + 1 == T;
                if (first && last)
                    ql_sweep<G, true, true>(tt, tabM, tabI, tabN, words, L, Lwave, active, sc, plane, xt, first_iter, dirty, o);
                else if (first)
                    ql_sweep<G, true, false>(tt, tabM, tabI, tabN, words, L, Lwave, active, sc, plane, xt, first_iter, dirty, o);
                else if (last)
                    ql_sweep<G, false, true>(tt, tabM, tabI, tabN, words, L, Lwave, active, sc, plane, xt, first_iter, dirty, o);
                else
                    ql_sweep<G, false, false>(tt, tabM, tabI, tabN, words, L, Lwave, active, sc, plane, xt, first_iter, dirty, o);
            }
            first_iter = false;
            if (!__syncthreads_or(dirty ? 1 : 0)) break;
        }

        if (has)
        {
            float const alt = fmaxf(o.E + xt.ET, o.C + xt.CT);
            float const nul = o.Rn;
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
    }
}

template <int G> static void launch_ql(dcp_qlane_args const *a, unsigned nblocks, hipStream_t s)
{
    hipLaunchKernelGGL((viterbi_qlane_kernel<G>), dim3(nblocks), dim3(256), 0, s, *a);
}

extern "C" int dcp_launch_qlane(int G, dcp_qlane_args const *a, unsigned nblocks, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    switch (G)
    {
    case 2: launch_ql<2>(a, nblocks, s); return 0;
    case 3: launch_ql<3>(a, nblocks, s); return 0;
    default: return -1;
    }
}
