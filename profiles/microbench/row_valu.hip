// Microbenchmark (round 3): the VALU-only cost of one query-lane Viterbi row (8 nodes per lane), registers
// only -- no LDS gathers, no boundary traffic -- for the row body with scalar transition adds (SGPR operand)
// and with the transition adds packed two per instruction (v_pk_add_f32 + SGPR pairs).  Two wavefronts per
// SIMD, one block per CU, as the production kernel runs.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -w -ffp-contract=off row_valu.hip -o row_valu && ./row_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float const __attribute__((address_space(4))) cfloat;
typedef f2 const __attribute__((address_space(4))) cf2;

__device__ __forceinline__ float mx3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float mx5(float a, float b, float c, float d, float e) { return mx3(mx3(a, b, c), d, e); }

struct Stamp
{
    unsigned long long t0, t1, r0, r1;
};

constexpr int KT = 8;

// VARIANT 0: scalar transition adds (SGPR operand); 1: packed (SGPR pairs); 2: scalar adds with the
// transitions in VGPRs (what it would cost if there were registers for them); 3 / 4: variant 0's operations
// issued in GROUPS -- all emission adds of 2 (3) or 4 (4) nodes, then their max3s, then their transition adds,
// then the remaining maxima, a scheduling barrier between the groups -- to see whether long runs of the
// 2-cycle class pair better across the two wavefronts of a SIMD
template <int VARIANT> __global__ __launch_bounds__(512, 2) void row_kernel(float *out, float const *trans, int rows, Stamp *st)
{
    extern __shared__ float lds[];
    float P[5][KT], Q[5][KT];
    float const x = threadIdx.x * 1e-3f;
#pragma unroll
    for (int h = 0; h < 5; ++h)
#pragma unroll
        for (int k = 0; k < KT; ++k)
            P[h][k] = -1.0f - x - h - k, Q[h][k] = -2.0f - x - h * 0.5f - k;
    // transitions: wave-uniform loads through the constant address space -> SGPRs
    cfloat *tt = (cfloat *)(unsigned long long)trans;
    cf2 *t2 = (cf2 *)tt;
    float ent[KT], mi[KT], ii[KT], mm[KT + 1], im[KT + 1], dm[KT + 1], md[KT + 1], dd[KT + 1];
    f2 ent2[4], mi2[4], ii2[4], mm2[4], im2[4], dm2[4], md2[4];
    float ddn[KT];
    if constexpr (VARIANT == 1)
    {
#pragma unroll
        for (int b = 0; b < 4; ++b)
            mi2[b] = t2[b], ii2[b] = t2[4 + b], ent2[b] = t2[8 + b], mm2[b] = t2[12 + b], im2[b] = t2[16 + b],
            dm2[b] = t2[20 + b], md2[b] = t2[24 + b];
#pragma unroll
        for (int k = 0; k < KT; ++k)
            ddn[k] = tt[56 + k];
    }
    else
    {
#pragma unroll
        for (int k = 0; k < KT; ++k)
            ent[k] = tt[k], mi[k] = tt[8 + k], ii[k] = tt[16 + k];
#pragma unroll
        for (int k = 0; k <= KT; ++k)
            mm[k] = tt[24 + k], im[k] = tt[33 + k], dm[k] = tt[42 + k], md[k] = tt[51 + k], dd[k] = tt[60 + k];
        if constexpr (VARIANT == 2)
        {
            // force the transitions into VGPRs (a lane-dependent zero is added)
            float const z = (threadIdx.x == 12345) ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < KT; ++k)
                ent[k] += z, mi[k] += z, ii[k] += z;
#pragma unroll
            for (int k = 0; k <= KT; ++k)
                mm[k] += z, im[k] += z, dm[k] += z, md[k] += z, dd[k] += z;
        }
    }
    float e[5][KT], eI[5];
#pragma unroll
    for (int l = 0; l < 5; ++l)
    {
        eI[l] = -0.3f * l - x;
#pragma unroll
        for (int k = 0; k < KT; ++k)
            e[l][k] = -0.1f * (l + 1) - 0.01f * k - x;
    }
    float E = -1e30f, Xm = -3.0f - x, Xd = -4.0f - x, Bj = -0.5f - x, acc = 0.0f;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();

#define ROW(PH)                                                                                                     \
    {                                                                                                               \
        constexpr int s1 = (PH + 4) % 5, s2 = (PH + 3) % 5, s3 = (PH + 2) % 5, s4 = (PH + 1) % 5, s5 = PH;           \
        if constexpr (VARIANT == 3 || VARIANT == 4)                                                                 \
        {                                                                                                           \
            constexpr int GN = VARIANT == 3 ? 2 : 4;                                                                \
            float pm = -1e30f, pi = -1e30f, pd = -1e30f;                                                            \
            _Pragma("unroll") for (int g = 0; g < KT / GN; ++g)                                                      \
            {                                                                                                       \
                float am[GN][5], ai[GN][5], m[GN], iv[GN];                                                          \
                _Pragma("unroll") for (int n = 0; n < GN; ++n)                                                       \
                {                                                                                                   \
                    int const k = g * GN + n;                                                                       \
                    am[n][0] = P[s1][k] + e[0][k], am[n][1] = P[s2][k] + e[1][k], am[n][2] = P[s3][k] + e[2][k];    \
                    am[n][3] = P[s4][k] + e[3][k], am[n][4] = P[s5][k] + e[4][k];                                   \
                    ai[n][0] = Q[s1][k] + eI[0], ai[n][1] = Q[s2][k] + eI[1], ai[n][2] = Q[s3][k] + eI[2];          \
                    ai[n][3] = Q[s4][k] + eI[3], ai[n][4] = Q[s5][k] + eI[4];                                       \
                }                                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
                _Pragma("unroll") for (int n = 0; n < GN; ++n)                                                       \
                {                                                                                                   \
                    m[n] = mx5(am[n][0], am[n][1], am[n][2], am[n][3], am[n][4]);                                   \
                    iv[n] = mx5(ai[n][0], ai[n][1], ai[n][2], ai[n][3], ai[n][4]);                                  \
                }                                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
                _Pragma("unroll") for (int n = 0; n < GN; ++n)                                                       \
                {                                                                                                   \
                    int const k = g * GN + n;                                                                       \
                    float d, pin;                                                                                   \
                    if (k == 0) d = Xd, pin = Xm;                                                                   \
                    else                                                                                            \
                    {                                                                                               \
                        d = fmaxf(pm + md[k], pd + dd[k]);                                                          \
                        pin = mx3(pm + mm[k], pi + im[k], pd + dm[k]);                                              \
                    }                                                                                               \
                    E = mx3(E, m[n], d);                                                                            \
                    P[PH][k] = fmaxf(Bj + ent[k], pin);                                                             \
                    Q[PH][k] = fmaxf(m[n] + mi[k], iv[n] + ii[k]);                                                  \
                    pm = m[n], pi = iv[n], pd = d;                                                                  \
                }                                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
            }                                                                                                       \
            Xm = mx3(pm + mm[KT], pi + im[KT], pd + dm[KT]);                                                        \
            Xd = fmaxf(pm + md[KT], pd + dd[KT]);                                                                   \
        }                                                                                                           \
        else if constexpr (VARIANT != 1)                                                                            \
        {                                                                                                           \
            float pm = -1e30f, pi = -1e30f, pd = -1e30f;                                                            \
            _Pragma("unroll") for (int k = 0; k < KT; ++k)                                                           \
            {                                                                                                       \
                float const m = mx5(P[s1][k] + e[0][k], P[s2][k] + e[1][k], P[s3][k] + e[2][k], P[s4][k] + e[3][k], \
                                    P[s5][k] + e[4][k]);                                                            \
                float const iv = mx5(Q[s1][k] + eI[0], Q[s2][k] + eI[1], Q[s3][k] + eI[2], Q[s4][k] + eI[3],        \
                                     Q[s5][k] + eI[4]);                                                             \
                float d, pin;                                                                                       \
                if (k == 0) d = Xd, pin = Xm;                                                                       \
                else                                                                                                \
                {                                                                                                   \
                    d = fmaxf(pm + md[k], pd + dd[k]);                                                              \
                    pin = mx3(pm + mm[k], pi + im[k], pd + dm[k]);                                                  \
                }                                                                                                   \
                E = mx3(E, m, d);                                                                                   \
                P[PH][k] = fmaxf(Bj + ent[k], pin);                                                                 \
                Q[PH][k] = fmaxf(m + mi[k], iv + ii[k]);                                                            \
                pm = m, pi = iv, pd = d;                                                                            \
            }                                                                                                       \
            Xm = mx3(pm + mm[KT], pi + im[KT], pd + dm[KT]);                                                        \
            Xd = fmaxf(pm + md[KT], pd + dd[KT]);                                                                   \
        }                                                                                                           \
        else                                                                                                        \
        {                                                                                                           \
            float carry_pin = Xm, carry_d = Xd;                                                                     \
            f2 const Bj2 = f2{Bj, Bj};                                                                              \
            _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                            \
            {                                                                                                       \
                int const k0 = 2 * b, k1 = 2 * b + 1;                                                               \
                float const m0 = mx5(P[s1][k0] + e[0][k0], P[s2][k0] + e[1][k0], P[s3][k0] + e[2][k0],              \
                                     P[s4][k0] + e[3][k0], P[s5][k0] + e[4][k0]);                                   \
                float const m1 = mx5(P[s1][k1] + e[0][k1], P[s2][k1] + e[1][k1], P[s3][k1] + e[2][k1],              \
                                     P[s4][k1] + e[3][k1], P[s5][k1] + e[4][k1]);                                   \
                float const i0 = mx5(Q[s1][k0] + eI[0], Q[s2][k0] + eI[1], Q[s3][k0] + eI[2], Q[s4][k0] + eI[3],    \
                                     Q[s5][k0] + eI[4]);                                                            \
                float const i1 = mx5(Q[s1][k1] + eI[0], Q[s2][k1] + eI[1], Q[s3][k1] + eI[2], Q[s4][k1] + eI[3],    \
                                     Q[s5][k1] + eI[4]);                                                            \
                f2 const Mp = f2{m0, m1}, Ip = f2{i0, i1};                                                          \
                f2 const q1 = Mp + mi2[b], q2 = Ip + ii2[b];                                                        \
                Q[PH][k0] = fmaxf(q1.x, q2.x);                                                                      \
                Q[PH][k1] = fmaxf(q1.y, q2.y);                                                                      \
                f2 const mdp = Mp + md2[b];                                                                         \
                float const d0 = carry_d;                                                                           \
                float const d1 = fmaxf(mdp.x, d0 + ddn[k0]);                                                        \
                float const d2 = fmaxf(mdp.y, d1 + ddn[k1]);                                                        \
                f2 const Dp = f2{d0, d1};                                                                           \
                f2 const a1 = Mp + mm2[b], a2 = Ip + im2[b], a3 = Dp + dm2[b];                                      \
                float const pin1 = mx3(a1.x, a2.x, a3.x);                                                           \
                float const pin2 = mx3(a1.y, a2.y, a3.y);                                                           \
                f2 const c = Bj2 + ent2[b];                                                                         \
                P[PH][k0] = fmaxf(c.x, carry_pin);                                                                  \
                P[PH][k1] = fmaxf(c.y, pin1);                                                                       \
                E = mx3(E, m0, d0);                                                                                 \
                E = mx3(E, m1, d1);                                                                                 \
                carry_pin = pin2;                                                                                   \
                carry_d = d2;                                                                                       \
            }                                                                                                       \
            Xm = carry_pin;                                                                                         \
            Xd = carry_d;                                                                                           \
        }                                                                                                           \
        Bj = Bj * 0.999f - 0.001f; /* keeps rows distinct: 2 more VALU per row in every variant */                   \
        acc += E;                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
    for (int j = 0; j + 5 <= rows; j += 5)
    {
        ROW(1) ROW(2) ROW(3) ROW(4) ROW(0)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int h = 0; h < 5; ++h)
#pragma unroll
        for (int k = 0; k < KT; ++k)
            acc += P[h][k] + Q[h][k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + Xm + Xd;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 8 + (threadIdx.x >> 6)] = Stamp{t0, t1, r0, r1};
}

typedef void (*kern_t)(float *, float const *, int, Stamp *);

static void run(char const *name, kern_t k)
{
    int const rows = 20000;
    float *out, *trans;
    Stamp *st;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&trans, 128 * sizeof(float));
    (void)hipMalloc(&st, 256 * 8 * sizeof(Stamp));
    std::vector<float> h(128);
    for (int i = 0; i < 128; ++i)
        h[i] = -0.05f * (1 + i % 7);
    (void)hipMemcpy(trans, h.data(), 128 * sizeof(float), hipMemcpyHostToDevice);
    size_t const lds = 100 * 1024;
    (void)hipFuncSetAttribute((void const *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int threads : {256, 512})
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), lds, 0, out, trans, 1000, st);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), lds, 0, out, trans, rows, st);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<Stamp> s(256 * 8);
        (void)hipMemcpy(s.data(), st, s.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        int const nw = threads / 64;
        for (int b = 0; b < 256; ++b)
        {
            unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
            for (int w = 0; w < nw; ++w)
            {
                Stamp const &x = s[b * 8 + w];
                t0 = std::min(t0, x.t0), t1 = std::max(t1, x.t1), r0 = std::min(r0, x.r0), r1 = std::max(r1, x.r1);
            }
            cyc += (double)(t1 - t0), rt += (double)(r1 - r0);
        }
        cyc /= 256, rt /= 256;
        double const wps = threads / 256.0;
        printf("%-44s %d waves/SIMD: %8.3f ms  %.2f GHz  %7.1f SIMD cycles per wavefront-row  -> VALU-only ceiling %6.0f Gcell/s at this clock\n",
               name, (int)wps, ms, cyc / rt * 0.1, cyc / (rows * wps), 256.0 * 4 * 64 * 8 / (cyc / (rows * wps)) * (cyc / rt * 0.1));
    }
    (void)hipFree(out), (void)hipFree(trans), (void)hipFree(st);
}

int main()
{
    run("scalar transition adds, SGPR operand", row_kernel<0>);
    run("packed transition adds, SGPR pairs", row_kernel<1>);
    run("scalar transition adds, VGPR operand", row_kernel<2>);
    run("variant 0 in groups of 2 nodes", row_kernel<3>);
    run("variant 0 in groups of 4 nodes", row_kernel<4>);
    return 0;
}
