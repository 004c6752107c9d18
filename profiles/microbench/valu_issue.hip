// Microbenchmark (round 3): SIMD cycles per VALU instruction on gfx950 with the SIMD saturated (2 and 4
// wavefronts per SIMD), for the instructions a max-plus Viterbi row is made of.  Decides (a) what the
// honest VALU ceiling of the query-lane kernel is and (b) whether packed float32 adds buy anything.
// Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -w valu_issue.hip -o valu_issue && ./valu_issue
// Method: every lane runs `iters` x 64 independent instructions (16 accumulators, each instruction
// depends only on the result 16 instructions back); one block per CU (100 KB of LDS), 256 x W threads
// = W wavefronts per SIMD.  Cycles = wall time x shader clock / (iters x 64 x W); the shader clock is
// read in the same launch as delta(s_memtime) / delta(s_memrealtime) x 100 MHz over the whole kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));

struct Stamp
{
    unsigned long long t0, t1, r0, r1;
};

#define KERNEL(NAME, ...)                                                                                 \
    __global__ __launch_bounds__(1024) void NAME(float *out, int iters, float s0f, Stamp *st)              \
    {                                                                                                       \
        extern __shared__ float lds[];                                                                      \
        float a[16];                                                                                        \
        f2 p[8];                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};                     \
        float sa = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s0f)));                   \
        f2 sp = f2{sa, sa};                                                                                 \
        float vb = s0f * (1.0f + (threadIdx.x & 63) * 1e-3f);                                               \
        f2 vp = f2{vb, vb * 1.5f};                                                                          \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                           \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int it = 0; it < iters; ++it)                                                                  \
        {                                                                                                   \
            __VA_ARGS__                                                                                        \
        }                                                                                                   \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                           \
        float acc = 0.0f;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) acc += a[i];                                         \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y;                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                   \
        if ((threadIdx.x & 63) == 0) st[blockIdx.x * 16 + (threadIdx.x >> 6)] = Stamp{t0, t1, r0, r1};      \
    }

#define S16(ASM, OUTS, ...)                                                                                \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) _Pragma("unroll") for (int i = 0; i < 16; ++i)             \
        asm volatile(ASM : OUTS : __VA_ARGS__);
#define P8(ASM, OUTS, ...)                                                                                 \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) _Pragma("unroll") for (int i = 0; i < 8; ++i)              \
        asm volatile(ASM : OUTS : __VA_ARGS__);

#define A_ "+v"(a[i])
#define AN_ "v"(a[(i + 1) & 15])
#define AN2_ "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15])

KERNEL(k_add_vv, S16("v_add_f32 %0, %0, %1", A_, "v"(vb)))
KERNEL(k_add_e64, S16("v_add_f32_e64 %0, %0, %1", A_, "v"(vb)))
KERNEL(k_sub_vv, S16("v_sub_f32 %0, %0, %1", A_, "v"(vb)))
KERNEL(k_mul_vv, S16("v_mul_f32 %0, %0, %1", A_, "v"(vb)))
KERNEL(k_fma_vv, S16("v_fma_f32 %0, %0, %1, %1", A_, "v"(vb)))
KERNEL(k_fmac_vv, S16("v_fmac_f32 %0, %1, %1", A_, "v"(vb)))
KERNEL(k_add_vs, S16("v_add_f32 %0, %1, %0", A_, "s"(sa)))
KERNEL(k_add_lit, S16("v_add_f32 %0, 0x3a83126f, %0", A_, "v"(vb)))
KERNEL(k_add_inl, S16("v_add_f32 %0, 1.0, %0", A_, "v"(vb)))
KERNEL(k_max_vv, S16("v_max_f32 %0, %0, %1", A_, AN_))
KERNEL(k_min_vv, S16("v_min_f32 %0, %0, %1", A_, AN_))
KERNEL(k_max3, S16("v_max3_f32 %0, %0, %1, %2", A_, AN2_))
KERNEL(k_max3_s, S16("v_max3_f32 %0, %0, %1, %2", A_, "v"(a[(i + 1) & 15]), "s"(sa)))
KERNEL(k_med3, S16("v_med3_f32 %0, %0, %1, %2", A_, AN2_))
KERNEL(k_pk_add_vv, P8("v_pk_add_f32 %0, %0, %1", "+v"(p[i]), "v"(vp)))
KERNEL(k_pk_add_vs, P8("v_pk_add_f32 %0, %0, %1", "+v"(p[i]), "s"(sp)))
KERNEL(k_pk_mul_vv, P8("v_pk_mul_f32 %0, %0, %1", "+v"(p[i]), "v"(vp)))
KERNEL(k_pk_fma_vv, P8("v_pk_fma_f32 %0, %0, %1, %1", "+v"(p[i]), "v"(vp)))
KERNEL(k_pk_mov, P8("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]", "+v"(p[i]), "v"(vp)))
KERNEL(k_add_u32, S16("v_add_u32 %0, %0, %1", A_, AN_))
KERNEL(k_and_b32, S16("v_and_b32 %0, %0, %1", A_, AN_))
KERNEL(k_max_i32, S16("v_max_i32 %0, %0, %1", A_, AN_))
KERNEL(k_min_u32, S16("v_min_u32 %0, %0, %1", A_, AN_))
KERNEL(k_max3_i32, S16("v_max3_i32 %0, %0, %1, %2", A_, AN2_))
KERNEL(k_lshl, S16("v_lshlrev_b32 %0, 1, %0", A_, "v"(vb)))
KERNEL(k_and_or, S16("v_and_or_b32 %0, %0, %1, %2", A_, AN2_))
KERNEL(k_bfe, S16("v_bfe_u32 %0, %0, 2, 10", A_, "v"(vb)))
KERNEL(k_mov, S16("v_mov_b32 %0, %1", "=v"(a[i]), AN_))
KERNEL(k_cndmask, S16("v_cndmask_b32 %0, %0, %1, vcc", A_, AN_))
KERNEL(k_cmp, S16("v_cmp_gt_f32 vcc, %0, %1", A_, AN_))
KERNEL(k_add_dpp, S16("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", A_, AN_))
KERNEL(k_pk_max_f16, S16("v_pk_max_f16 %0, %0, %1", A_, AN_))
// the M_k / I_k shape of one node-row: 5 adds + 2 max3 (7 instructions, 7 lane-ops) ...
KERNEL(k_mk_plain, _Pragma("unroll") for (int n = 0; n < 8; ++n) {
    float t0_, t1_, t2_, t3_, t4_;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t0_) : "v"(a[n]), "v"(a[(n + 1) & 15]));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t1_) : "v"(a[n]), "v"(a[(n + 2) & 15]));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t2_) : "v"(a[n]), "v"(a[(n + 3) & 15]));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t3_) : "v"(a[n]), "v"(a[(n + 4) & 15]));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t4_) : "v"(a[n]), "v"(a[(n + 5) & 15]));
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(t0_) : "v"(t1_), "v"(t2_));
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a[n + 8]) : "v"(t0_), "v"(t3_), "v"(t4_));
})
// ... and for two nodes at once with packed adds: 5 pk_add + 4 max3 (9 instructions, 14 lane-ops)
KERNEL(k_mk_packed, _Pragma("unroll") for (int n = 0; n < 4; ++n) {
    f2 t0_, t1_, t2_, t3_, t4_;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t0_) : "v"(p[n]), "v"(p[(n + 1) & 7]));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t1_) : "v"(p[n]), "v"(p[(n + 2) & 7]));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t2_) : "v"(p[n]), "v"(p[(n + 3) & 7]));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t3_) : "v"(p[n]), "v"(p[(n + 4) & 7]));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t4_) : "v"(p[n]), "v"(p[(n + 5) & 7]));
    float x0 = t0_.x, y0 = t0_.y;
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(t1_.x), "v"(t2_.x));
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(y0) : "v"(t1_.y), "v"(t2_.y));
    float x1, y1;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(x0), "v"(t3_.x), "v"(t4_.x));
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(y1) : "v"(y0), "v"(t3_.y), "v"(t4_.y));
    p[n + 4] = f2{x1, y1};
})

typedef void (*kern_t)(float *, int, float, Stamp *);

static void run(char const *name, kern_t k, int instr_per_iter, int ops_per_iter)
{
    int const iters = 20000;
    float *out;
    Stamp *st;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    (void)hipMalloc(&st, 256 * 16 * sizeof(Stamp));
    printf("%-22s", name);
    for (int wps = 1; wps <= 4; wps *= 2)
    {
        int const threads = 256 * wps;
        size_t const lds = 100 * 1024; // one block per CU
        (void)hipFuncSetAttribute((void const *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), lds, 0, out, 2000, 1e-3f, st);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), lds, 0, out, iters, 1e-3f, st);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<Stamp> h(256 * 16);
        (void)hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
        // per block: the span from the first wavefront's start to the last one's end, in shader cycles and in 100 MHz ticks
        double cyc = 0, rt = 0;
        for (int b = 0; b < 256; ++b)
        {
            unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
            for (int w = 0; w < 4 * wps; ++w)
            {
                Stamp const &s = h[b * 16 + w];
                t0 = std::min(t0, s.t0), t1 = std::max(t1, s.t1), r0 = std::min(r0, s.r0), r1 = std::max(r1, s.r1);
            }
            cyc += (double)(t1 - t0), rt += (double)(r1 - r0);
        }
        cyc /= 256.0, rt /= 256.0;
        double const ghz = cyc / rt * 0.1; // s_memrealtime: 100 MHz
        double const cyc_per_instr = cyc / ((double)iters * instr_per_iter * wps);
        printf("  W=%d: %6.3f ms %5.2f GHz %6.3f cyc/instr %6.3f cyc/op |", wps, ms, ghz, cyc_per_instr,
               cyc / ((double)iters * ops_per_iter * wps));
    }
    printf("\n");
    (void)hipFree(out);
    (void)hipFree(st);
}

int main()
{
    printf("cycles are SIMD cycles per wavefront-instruction with W wavefronts per SIMD sharing it (block span / instructions issued per SIMD)\n");
#define RUN(k) run(#k, k, 64, 64)
    RUN(k_add_vv);
    RUN(k_add_e64);
    RUN(k_sub_vv);
    RUN(k_mul_vv);
    RUN(k_fma_vv);
    RUN(k_fmac_vv);
    RUN(k_add_vs);
    RUN(k_add_lit);
    RUN(k_add_inl);
    RUN(k_max_vv);
    RUN(k_min_vv);
    RUN(k_max3);
    RUN(k_max3_s);
    RUN(k_med3);
    run("k_pk_add_vv", k_pk_add_vv, 64, 128);
    run("k_pk_add_vs", k_pk_add_vs, 64, 128);
    run("k_pk_mul_vv", k_pk_mul_vv, 64, 128);
    run("k_pk_fma_vv", k_pk_fma_vv, 64, 128);
    run("k_pk_mov", k_pk_mov, 64, 128);
    RUN(k_add_u32);
    RUN(k_and_b32);
    RUN(k_max_i32);
    RUN(k_min_u32);
    RUN(k_max3_i32);
    RUN(k_lshl);
    RUN(k_and_or);
    RUN(k_bfe);
    RUN(k_mov);
    RUN(k_cndmask);
    RUN(k_cmp);
    RUN(k_add_dpp);
    RUN(k_pk_max_f16);
    run("k_mk_plain 5add+2max3", k_mk_plain, 56, 56);
    run("k_mk_packed 5pk+4max3", k_mk_packed, 36, 56);
    return 0;
}
