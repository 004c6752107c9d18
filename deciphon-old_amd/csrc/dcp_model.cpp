// dcp_model.cpp -- host side of the scan engine: builds the compact profile
// ("dcpx") the device consumes. Product code; never calls the oracle.
//
// Restates, MI355X-first, what the reference's model layer computes:
//   protein_model_init / add_node / add_trans / setup_transitions
//       src/model/protein_model.c:49-137, 460-500
//   setup_nuclt_dist / codon_lprob / nuclt_lprob      :342-408
//   calculate_occupancy, setup_entry_trans            :258-283, :410-439
//   protein_profile_sample / setup                    src/model/protein_profile.c:155-304
// Differences by design: the reference keeps per-state imm_dp emission tables
// on the host and re-reads them per pair; here only the 129-float nuclt_dist
// per node and an 8-row transition matrix are kept, and all arithmetic is done
// in double in the probability domain and rounded once to float32 (imm_float).
#include "dcp_host.h"

#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

namespace
{
constexpr double kNegInf = -std::numeric_limits<double>::infinity();
constexpr float kNegInfF = -std::numeric_limits<float>::infinity();

// --- imm_rnd: xoshiro256+ seeded with splitmix64 (pinned by the reference's
// golden test/protein_profile.c:41 through the oracle's search) -------------
struct Rnd
{
    uint64_t s[4];
    explicit Rnd(uint64_t seed)
    {
        for (auto &v : s)
        {
            uint64_t z = (seed += 0x9e3779b97f4a7c15ULL);
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            v = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    double next()
    {
        uint64_t const r = s[0] + s[3];
        uint64_t const t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return (double)(r >> 11) * 0x1.0p-53;
    }
};

double logsumexp(double const *v, int n)
{
    double m = kNegInf;
    for (int i = 0; i < n; ++i)
        m = std::fmax(m, v[i]);
    if (!(m > kNegInf)) return kNegInf;
    double s = 0;
    for (int i = 0; i < n; ++i)
        s += std::exp(v[i] - m);
    return m + std::log(s);
}

double logadd(double a, double b)
{
    double v[2] = {a, b};
    return logsumexp(v, 2);
}

// imm_lprob_sample + imm_lprob_normalize on imm_float values
void sample_normalized(Rnd &rnd, int n, float *out, int const *force_zero,
                       int nforce)
{
    std::vector<double> lp((size_t)n);
    for (int i = 0; i < n; ++i)
        lp[(size_t)i] = (double)(float)std::log(rnd.next());
    for (int i = 0; i < nforce; ++i)
        lp[(size_t)force_zero[i]] = kNegInf;
    double z = logsumexp(lp.data(), n);
    for (int i = 0; i < n; ++i)
        out[i] = (float)(lp[(size_t)i] - z);
}

// NCBI genetic code 1 (imm_gc table 1), codon index = 16*b1 + 4*b2 + b3 in the
// TCAG order of the published table; converted to ACGT ids below.
constexpr char kGcAmino[] =
    "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
constexpr int kTcagToAcgt[4] = {3, 1, 0, 2};
constexpr char kAminoSymbols[] = "ACDEFGHIKLMNPQRSTVWY";

struct CodonTable
{
    char aa[64];        // amino acid (or '*') of codon a*16+b*4+c in ACGT ids
    int synonyms[128];  // number of codons per amino acid letter
    CodonTable()
    {
        std::memset(synonyms, 0, sizeof synonyms);
        for (int i = 0; i < 64; ++i)
        {
            int a = kTcagToAcgt[i >> 4], b = kTcagToAcgt[(i >> 2) & 3],
                c = kTcagToAcgt[i & 3];
            aa[a * 16 + b * 4 + c] = kGcAmino[i];
            synonyms[(int)kGcAmino[i]]++;
        }
    }
};
CodonTable const g_codons;

// setup_nuclt_dist: amino log-(odds|probs) -> 4 base lprobs + 125 codon marginals
void make_nuclt_dist(float const aa_lprobs[20], float out[DCP_NDIST])
{
    double by_letter[128];
    for (double &v : by_letter)
        v = kNegInf;
    for (int i = 0; i < 20; ++i)
    {
        int letter = kAminoSymbols[i];
        by_letter[letter] =
            (double)aa_lprobs[i] - std::log((double)g_codons.synonyms[letter]);
    }
    double codon_lp[64];
    for (int i = 0; i < 64; ++i)
        codon_lp[i] = by_letter[(int)g_codons.aa[i]]; // stops ('*') stay -inf
    double z = logsumexp(codon_lp, 64);
    double p[64]; // probability domain from here on
    for (int i = 0; i < 64; ++i)
        p[i] = std::exp(codon_lp[i] - z);

    double base[4] = {0, 0, 0, 0};
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
            for (int c = 0; c < 4; ++c)
            {
                double third = p[a * 16 + b * 4 + c] / 3.0;
                base[a] += third;
                base[b] += third;
                base[c] += third;
            }
    for (int i = 0; i < 4; ++i)
        out[i] = (float)std::log(base[i]);

    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b)
            for (int c = 0; c < 5; ++c)
            {
                double s = 0;
                for (int i = (a == 4 ? 0 : a); i < (a == 4 ? 4 : a + 1); ++i)
                    for (int j = (b == 4 ? 0 : b); j < (b == 4 ? 4 : b + 1); ++j)
                        for (int k = (c == 4 ? 0 : c); k < (c == 4 ? 4 : c + 1); ++k)
                            s += p[i * 16 + j * 4 + k];
                out[4 + a * 25 + b * 5 + c] = (float)std::log(s);
            }
}

} // namespace

struct dcp_profile
{
    char accession[DCP_PROFILE_ACC_SIZE];
    unsigned core_size;
    int entry_dist;
    float epsilon;
    std::vector<char> consensus;
    std::vector<float> trans8;     // [8][M]
    std::vector<float> match_dist; // [M][129]
    float null_dist[DCP_NDIST];
    float insert_dist[DCP_NDIST];
    // exp() of the base and codon probabilities of the null and insert distributions, for dcp_profile_decode:
    // most steps of a path are flank steps in N / C / J, all decoded against the null distribution, and the 68
    // exp() calls were two thirds of a decode.  Filled on first use (profiles are read-only once built).
    struct DecodeExp
    {
        double b[4], pc[64];
    };
    mutable std::once_flag decode_once;
    mutable DecodeExp decode_exp[2]; // 0 null, 1 insert
};

// calculate_occupancy + setup_entry_trans (protein_model.c:258-283,410-439)
static void entry_scores(unsigned M, int entry_dist, float const *trans,
                         float *entry)
{
    if (entry_dist == DCP_ENTRY_DIST_UNIFORM)
    {
        // imm_log(2.0 / (M * (M + 1))) * M  -- reproduced as written (:415)
        float Mf = (float)M;
        float cost = (float)(std::log(2.0 / ((double)Mf * ((double)Mf + 1.0))) * (double)Mf);
        for (unsigned i = 0; i < M; ++i)
            entry[i] = cost;
        return;
    }
    std::vector<double> locc(M);
    auto T = [&](unsigned i, int f) { return (double)trans[7 * i + f]; };
    enum { MM, MI, MD, IM, II, DM, DD };
    locc[0] = logadd(T(0, MI), T(0, MM));
    for (unsigned i = 1; i < M; ++i)
    {
        double v0 = locc[i - 1] + logadd(T(i, MM), T(i, MI));
        double v1 = std::log1p(-std::exp(locc[i - 1])) + T(i, DM);
        locc[i] = logadd(v0, v1);
    }
    double logZ = kNegInf;
    for (unsigned i = 0; i < M; ++i)
        logZ = logadd(logZ, locc[i] + std::log((double)(M - i)));
    for (unsigned i = 0; i < M; ++i)
        entry[i] = (float)(locc[i] - logZ);
}

extern "C" {

dcp_profile *dcp_profile_new(char const *accession, unsigned core_size,
                             int entry_dist, float epsilon,
                             float const *null_lprobs,
                             float const *match_lprobs, float const *trans,
                             char const *consensus, int *rc)
{
    auto fail = [&](int code) -> dcp_profile * {
        if (rc) *rc = code;
        return nullptr;
    };
    if (core_size == 0 || core_size > DCP_CORE_SIZE_MAX) return fail(DCP_EINVAL);
    if (!null_lprobs || !match_lprobs || !trans) return fail(DCP_EINVAL);
    if (entry_dist != DCP_ENTRY_DIST_UNIFORM && entry_dist != DCP_ENTRY_DIST_OCCUPANCY)
        return fail(DCP_EINVAL);
    if (!(epsilon >= 0.0f && epsilon <= 1.0f)) return fail(DCP_EINVAL); // protein_cfg.h:18

    dcp_profile *p = new (std::nothrow) dcp_profile();
    if (!p) return fail(DCP_ENOMEM);
    unsigned const M = core_size;
    std::memset(p->accession, 0, sizeof p->accession);
    if (accession) std::strncpy(p->accession, accession, sizeof p->accession - 1);
    p->core_size = M;
    p->entry_dist = entry_dist;
    p->epsilon = epsilon;
    p->consensus.assign(M + 1, '\0');
    bool ended = !consensus; // a consensus shorter than the core is padded, never read past its NUL
    for (unsigned i = 0; i < M; ++i)
    {
        if (!ended && consensus[i] == '\0') ended = true;
        p->consensus[i] = ended ? '-' : consensus[i];
    }

    // protein_model_init: null dist from the null amino lprobs, insert dist from
    // all-zero log-odds (protein_model.c:122-127)
    make_nuclt_dist(null_lprobs, p->null_dist);
    float const zeros[20] = {0};
    make_nuclt_dist(zeros, p->insert_dist);
    // protein_model_add_node: match dist from lodds = lprob - null (:60-66)
    p->match_dist.resize((size_t)M * DCP_NDIST);
    for (unsigned k = 0; k < M; ++k)
    {
        float lodds[20];
        for (int i = 0; i < 20; ++i)
            lodds[i] = match_lprobs[k * 20 + i] - null_lprobs[i];
        make_nuclt_dist(lodds, &p->match_dist[(size_t)k * DCP_NDIST]);
    }

    // setup_transitions (:460-500) folded into per-destination-node rows.
    // trans[j] (j = i+1) carries the edges between node i and node i+1 and the
    // MI/II edges of node i (:469-489).  B->M1 = trans[0].MM and M_M->E =
    // trans[M].MM are overwritten by entry / exit scores (:421,:433,:448).
    enum { MM, MI, MD, IM, II, DM, DD };
    p->trans8.assign((size_t)8 * M, kNegInfF);
    float *t8 = p->trans8.data();
    entry_scores(M, entry_dist, trans, t8 + DCP_T_ENTRY * M);
    for (unsigned i = 0; i + 1 < M; ++i)
    {
        float const *t = trans + 7 * (i + 1);
        t8[DCP_T_MI * M + i] = t[MI];
        t8[DCP_T_II * M + i] = t[II];
        t8[DCP_T_MM * M + i + 1] = t[MM];
        t8[DCP_T_IM * M + i + 1] = t[IM];
        t8[DCP_T_MD * M + i + 1] = t[MD];
        t8[DCP_T_DD * M + i + 1] = t[DD];
        t8[DCP_T_DM * M + i + 1] = t[DM];
    }
    if (rc) *rc = DCP_OK;
    return p;
}

dcp_profile *dcp_profile_sample(char const *accession, unsigned seed,
                                unsigned core_size, int entry_dist,
                                float epsilon, int *rc)
{
    if (core_size < 2 || core_size > DCP_CORE_SIZE_MAX) // assert(core_size >= 2) :262
    {
        if (rc) *rc = DCP_EINVAL;
        return nullptr;
    }
    Rnd rnd(seed);
    unsigned const M = core_size;
    float null_lp[20];
    sample_normalized(rnd, 20, null_lp, nullptr, 0);
    std::vector<float> match((size_t)20 * M), trans((size_t)7 * (M + 1));
    for (unsigned k = 0; k < M; ++k)
        sample_normalized(rnd, 20, &match[(size_t)20 * k], nullptr, 0);
    for (unsigned i = 0; i <= M; ++i)
    {
        int zero[2] = {6 /*DD*/, 2 /*MD*/};
        int nz = i == 0 ? 1 : (i == M ? 2 : 0);
        sample_normalized(rnd, 7, &trans[(size_t)7 * i], zero, nz);
    }
    return dcp_profile_new(accession, M, entry_dist, epsilon, null_lp,
                           match.data(), trans.data(), nullptr, rc);
}

dcp_profile *dcp_profile_from_parts(char const *accession, unsigned core_size, int entry_dist, float epsilon,
                                    char const *consensus, float const *trans8, float const *null_dist,
                                    float const *insert_dist, float const *match_dist, int *rc)
{
    auto fail = [&](int code) -> dcp_profile * {
        if (rc) *rc = code;
        return nullptr;
    };
    if (core_size == 0 || core_size > DCP_CORE_SIZE_MAX) return fail(DCP_EINVAL);
    if (!trans8 || !null_dist || !insert_dist || !match_dist) return fail(DCP_EINVAL);
    if (!(epsilon >= 0.0f && epsilon <= 1.0f)) return fail(DCP_EINVAL);
    dcp_profile *p = new (std::nothrow) dcp_profile();
    if (!p) return fail(DCP_ENOMEM);
    unsigned const M = core_size;
    std::memset(p->accession, 0, sizeof p->accession);
    if (accession) std::strncpy(p->accession, accession, sizeof p->accession - 1);
    p->core_size = M;
    p->entry_dist = entry_dist;
    p->epsilon = epsilon;
    p->consensus.assign(M + 1, '\0');
    bool ended = !consensus;
    for (unsigned i = 0; i < M; ++i)
    {
        if (!ended && consensus[i] == '\0') ended = true;
        p->consensus[i] = ended ? '-' : consensus[i];
    }
    p->trans8.assign(trans8, trans8 + (size_t)8 * M);
    p->match_dist.assign(match_dist, match_dist + (size_t)M * DCP_NDIST);
    std::memcpy(p->null_dist, null_dist, sizeof p->null_dist);
    std::memcpy(p->insert_dist, insert_dist, sizeof p->insert_dist);
    // a NaN anywhere would poison every max: reject it here rather than on the device
    for (float v : p->trans8)
        if (v != v) { delete p; return fail(DCP_EINVAL); }
    for (float v : p->match_dist)
        if (v != v) { delete p; return fail(DCP_EINVAL); }
    for (int i = 0; i < DCP_NDIST; ++i)
        if (p->null_dist[i] != p->null_dist[i] || p->insert_dist[i] != p->insert_dist[i]) { delete p; return fail(DCP_EINVAL); }
    if (rc) *rc = DCP_OK;
    return p;
}

void dcp_rnd_seed(uint64_t state[4], uint64_t seed)
{
    Rnd r(seed);
    std::memcpy(state, r.s, sizeof r.s);
}

double dcp_rnd_next(uint64_t state[4])
{
    Rnd r(0);
    std::memcpy(r.s, state, sizeof r.s);
    double const v = r.next();
    std::memcpy(state, r.s, sizeof r.s);
    return v;
}

void dcp_lprob_normalize(unsigned n, float *lprobs)
{
    std::vector<double> lp(lprobs, lprobs + n);
    double const z = logsumexp(lp.data(), (int)n);
    for (unsigned i = 0; i < n; ++i)
        lprobs[i] = (float)(lp[i] - z);
}

int dcp_profile_entry_dist(dcp_profile const *p) { return p->entry_dist; }

void dcp_profile_del(dcp_profile *p) { delete p; }
unsigned dcp_profile_core_size(dcp_profile const *p) { return p->core_size; }
char const *dcp_profile_accession(dcp_profile const *p) { return p->accession; }
char const *dcp_profile_consensus(dcp_profile const *p) { return p->consensus.data(); }
float const *dcp_profile_trans8(dcp_profile const *p) { return p->trans8.data(); }
float const *dcp_profile_null_dist(dcp_profile const *p) { return p->null_dist; }
float const *dcp_profile_insert_dist(dcp_profile const *p) { return p->insert_dist; }
float const *dcp_profile_match_dist(dcp_profile const *p) { return p->match_dist.data(); }
float dcp_profile_epsilon(dcp_profile const *p) { return p->epsilon; }

// Frame-state emission in the probability domain (imm frame state; the 4-event
// indel model of SURVEY Appendix A).  b = base probs, C = codon marginals.
void dcp_frame_table_host(float const dist[DCP_NDIST], float epsilon,
                          float out[DCP_NCODES])
{
    double b[4], C[125];
    for (int i = 0; i < 4; ++i)
        b[i] = std::exp((double)dist[i]);
    for (int i = 0; i < 125; ++i)
        C[i] = std::exp((double)dist[4 + i]);
    double const e = (double)epsilon, f = 1.0 - (double)epsilon;
    double const e2 = e * e, f2 = f * f;
    auto c3 = [&](int x, int y, int z) { return C[x * 25 + y * 5 + z]; };
    auto s1 = [&](int x) { return c3(x, 4, 4) + c3(4, x, 4) + c3(4, 4, x); };
    auto s2 = [&](int x, int y) { return c3(4, x, y) + c3(x, 4, y) + c3(x, y, 4); };

    for (int x = 0; x < 4; ++x)
        out[x] = (float)std::log(e2 * f2 / 3.0 * s1(x));
    for (int v = 0; v < 16; ++v)
    {
        int x1 = v >> 2, x2 = v & 3;
        double p = 2.0 * e * f2 * f / 3.0 * s2(x1, x2) +
                   e2 * e * f / 3.0 * (b[x2] * s1(x1) + b[x1] * s1(x2));
        out[4 + v] = (float)std::log(p);
    }
    for (int v = 0; v < 64; ++v)
    {
        int x1 = v >> 4, x2 = (v >> 2) & 3, x3 = v & 3;
        double p = f2 * f2 * c3(x1, x2, x3) +
                   4.0 * e2 * f2 / 9.0 *
                       (b[x1] * s2(x2, x3) + b[x2] * s2(x1, x3) + b[x3] * s2(x1, x2)) +
                   e2 * e2 / 9.0 *
                       (b[x1] * b[x2] * s1(x3) + b[x1] * b[x3] * s1(x2) +
                        b[x2] * b[x3] * s1(x1));
        out[20 + v] = (float)std::log(p);
    }
    for (int v = 0; v < 256; ++v)
    {
        int x[4] = {v >> 6, (v >> 4) & 3, (v >> 2) & 3, v & 3};
        double one = b[x[0]] * c3(x[1], x[2], x[3]) + b[x[1]] * c3(x[0], x[2], x[3]) +
                     b[x[2]] * c3(x[0], x[1], x[3]) + b[x[3]] * c3(x[0], x[1], x[2]);
        double two = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j)
            {
                int r[2], n = 0;
                for (int k = 0; k < 4; ++k)
                    if (k != i && k != j) r[n++] = x[k];
                two += b[x[i]] * b[x[j]] * s2(r[0], r[1]);
            }
        out[84 + v] = (float)std::log(e * f2 * f / 2.0 * one + e2 * e * f / 9.0 * two);
    }
    for (int v = 0; v < 1024; ++v)
    {
        int x[5] = {v >> 8, (v >> 6) & 3, (v >> 4) & 3, (v >> 2) & 3, v & 3};
        double s = 0;
        for (int i = 0; i < 5; ++i)
            for (int j = i + 1; j < 5; ++j)
            {
                int r[3], n = 0;
                for (int k = 0; k < 5; ++k)
                    if (k != i && k != j) r[n++] = x[k];
                s += b[x[i]] * b[x[j]] * c3(r[0], r[1], r[2]);
            }
        out[340 + v] = (float)std::log(e2 * f2 / 10.0 * s);
    }
}

// protein_profile_setup (protein_profile.c:155-216)
int dcp_xtrans(unsigned seq_size, int multi_hits, int hmmer3_compat,
               float out[DCP_NXTRANS])
{
    if (seq_size == 0) return DCP_EINVAL;
    float const L = (float)seq_size;
    float q = 0.0f, log_q = kNegInfF;
    if (multi_hits)
    {
        q = 0.5f;
        log_q = (float)std::log(0.5);
    }
    float lp = (float)std::log((double)L) - (float)std::log((double)(L + 2 + q / (1 - q)));
    float l1p = (float)std::log((double)(2 + q / (1 - q))) -
                (float)std::log((double)(L + 2 + q / (1 - q)));
    float lr = (float)std::log((double)L) - (float)std::log((double)(L + 1));
    float NN = lp, CC = lp, JJ = lp, NB = l1p, CT = l1p, JB = l1p, RR = lr;
    float EJ = log_q, EC = (float)std::log((double)(1 - q));
    if (hmmer3_compat) NN = CC = JJ = 0.0f;
    out[DCP_X_RR] = RR;
    out[DCP_X_SB] = NB;
    out[DCP_X_SN] = NN;
    out[DCP_X_NN] = NN;
    out[DCP_X_NB] = NB;
    out[DCP_X_ET] = EC + CT;
    out[DCP_X_EC] = EC + CC;
    out[DCP_X_CC] = CC;
    out[DCP_X_CT] = CT;
    out[DCP_X_EB] = EJ + JB;
    out[DCP_X_EJ] = EJ + JJ;
    out[DCP_X_JJ] = JJ;
    out[DCP_X_JB] = JB;
    return DCP_OK;
}

float dcp_lrt(float null_loglik, float alt_loglik)
{
    return -2 * (null_loglik - alt_loglik);
}

// xmath_partition_size (xmath.h:24-30) + partition_it (profile_reader.c:54-72)
unsigned dcp_partition_by_count(unsigned nprofiles, unsigned npartitions,
                                unsigned part_size[DCP_NUM_THREADS])
{
    if (npartitions == 0 || npartitions > DCP_NUM_THREADS) return 0;
    unsigned nparts = npartitions < nprofiles ? npartitions : nprofiles;
    for (unsigned i = 0; i < DCP_NUM_THREADS; ++i)
        part_size[i] = 0;
    if (nparts == 0) return 0;
    unsigned i = 0, size = 0;
    unsigned const chunk = (nprofiles + nparts - 1) / nparts;
    for (unsigned j = 0; j < nprofiles; ++j)
    {
        unsigned want = chunk * i <= nprofiles
                            ? (chunk < nprofiles - chunk * i ? chunk : nprofiles - chunk * i)
                            : 0;
        if (++size >= want)
        {
            part_size[i] = size;
            ++i;
            size = 0;
            if (i >= nparts) break;
        }
    }
    return nparts;
}

void dcp_partition_by_cells(unsigned const *core_sizes, unsigned nprofiles,
                            unsigned npartitions, unsigned *part_begin)
{
    uint64_t total = 0;
    for (unsigned i = 0; i < nprofiles; ++i)
        total += core_sizes[i];
    part_begin[0] = 0;
    uint64_t acc = 0;
    unsigned p = 0;
    for (unsigned g = 1; g < npartitions; ++g)
    {
        // boundary g: first profile whose prefix sum reaches g/npartitions of the work
        uint64_t target = (total * g + npartitions / 2) / npartitions;
        while (p < nprofiles && acc + core_sizes[p] / 2 < target)
            acc += core_sizes[p++];
        part_begin[g] = p;
    }
    part_begin[npartitions] = nprofiles;
}

} // extern "C"

// ---------------------------------------------------------------------------
// Hit post-processing: state names, codon decode, product rows
// ---------------------------------------------------------------------------
namespace
{
// imm frame state: probability of emitting word x (1..5 bases) given base probs b and
// codon marginals C (5x5x5, index 4 = wildcard) -- the formula of dcp_frame_table_host,
// evaluated for one word.
// C3(p, q, r): codon marginal with wildcard index 4 -- a table (below) or computed on the fly (decode)
template <class C3> double frame_prob_word_of(double const *b, C3 const &c3, double e, double f, uint8_t const *x, unsigned len)
{
    auto s1 = [&](int p) { return c3(p, 4, 4) + c3(4, p, 4) + c3(4, 4, p); };
    auto s2 = [&](int p, int q) { return c3(4, p, q) + c3(p, 4, q) + c3(p, q, 4); };
    double const e2 = e * e, f2 = f * f;
    switch (len)
    {
    case 1: return e2 * f2 / 3.0 * s1(x[0]);
    case 2:
        return 2.0 * e * f2 * f / 3.0 * s2(x[0], x[1]) +
               e2 * e * f / 3.0 * (b[x[1]] * s1(x[0]) + b[x[0]] * s1(x[1]));
    case 3:
        return f2 * f2 * c3(x[0], x[1], x[2]) +
               4.0 * e2 * f2 / 9.0 * (b[x[0]] * s2(x[1], x[2]) + b[x[1]] * s2(x[0], x[2]) + b[x[2]] * s2(x[0], x[1])) +
               e2 * e2 / 9.0 * (b[x[0]] * b[x[1]] * s1(x[2]) + b[x[0]] * b[x[2]] * s1(x[1]) + b[x[1]] * b[x[2]] * s1(x[0]));
    case 4:
    {
        double one = b[x[0]] * c3(x[1], x[2], x[3]) + b[x[1]] * c3(x[0], x[2], x[3]) +
                     b[x[2]] * c3(x[0], x[1], x[3]) + b[x[3]] * c3(x[0], x[1], x[2]);
        double two = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j)
            {
                int r[2], n = 0;
                for (int k = 0; k < 4; ++k)
                    if (k != i && k != j) r[n++] = x[k];
                two += b[x[i]] * b[x[j]] * s2(r[0], r[1]);
            }
        return e * f2 * f / 2.0 * one + e2 * e * f / 9.0 * two;
    }
    case 5:
    {
        double s = 0;
        for (int i = 0; i < 5; ++i)
            for (int j = i + 1; j < 5; ++j)
            {
                int r[3], n = 0;
                for (int k = 0; k < 5; ++k)
                    if (k != i && k != j) r[n++] = x[k];
                s += b[x[i]] * b[x[j]] * c3(r[0], r[1], r[2]);
            }
        return e2 * f2 / 10.0 * s;
    }
    default: return std::numeric_limits<double>::quiet_NaN();
    }
}
} // namespace

extern "C" {

// protein_state_name: src/model/protein_state.c:5-39
unsigned dcp_state_name(unsigned id, char name[8])
{
    unsigned const msb = id & (3u << 14);
    if (msb == (3u << 14))
    {
        static char const ext[] = "RSNBEJCT";
        unsigned i = id & 0x3FFFu;
        name[0] = i < 8 ? ext[i] : '?';
        name[1] = '\0';
        return 1;
    }
    name[0] = msb == 0 ? 'M' : msb == (1u << 14) ? 'I' : 'D';
    return (unsigned)snprintf(name + 1, 7, "%u", id & 0x3FFFu) + 1;
}

char dcp_gc_decode(uint8_t const codon[3])
{
    if (codon[0] > 3 || codon[1] > 3 || codon[2] > 3) return 'X';
    return g_codons.aa[codon[0] * 16 + codon[1] * 4 + codon[2]];
}

// protein_profile_decode (src/model/protein_profile.c:306-331): the distribution is the
// insert dist for I states, the node's match dist for M states, the null dist otherwise;
// imm_frame_cond_decode = arg-max over the 64 codons of p(fragment, codon).
int dcp_profile_decode(dcp_profile const *p, uint8_t const *frag, unsigned len, unsigned state_id,
                       uint8_t codon[3])
{
    if (!p || !frag || !codon || len < 1 || len > 5) return DCP_EINVAL;
    for (unsigned i = 0; i < len; ++i)
        if (frag[i] > 3) return DCP_EINVAL;
    unsigned const msb = state_id & (3u << 14);
    float const *dist;
    if (msb == (1u << 14))
        dist = p->insert_dist;
    else if (msb == 0)
    {
        unsigned k = (state_id & 0x3FFFu) - 1u; // protein_state_idx
        if (k >= p->core_size) return DCP_EINVAL;
        dist = &p->match_dist[(size_t)k * DCP_NDIST];
    }
    else if (state_id == ((3u << 14) | 1u) || state_id == ((3u << 14) | 3u) ||
             state_id == ((3u << 14) | 4u) || state_id == ((3u << 14) | 7u) || msb == (2u << 14))
        return DCP_EINVAL; // mute states emit nothing: assert(!protein_state_is_mute) :310
    else
        dist = p->null_dist;
    auto const fill = [](float const *d, dcp_profile::DecodeExp &x) {
        for (int i = 0; i < 4; ++i)
            x.b[i] = std::exp((double)d[i]);
        for (int a = 0; a < 4; ++a)
            for (int bb = 0; bb < 4; ++bb)
                for (int cc = 0; cc < 4; ++cc)
                    x.pc[a * 16 + bb * 4 + cc] = std::exp((double)d[4 + a * 25 + bb * 5 + cc]);
    };
    dcp_profile::DecodeExp own;
    dcp_profile::DecodeExp const *x = &own;
    if (dist == p->null_dist || dist == p->insert_dist)
    {
        std::call_once(p->decode_once, [&]() { fill(p->null_dist, p->decode_exp[0]), fill(p->insert_dist, p->decode_exp[1]); });
        x = &p->decode_exp[dist == p->insert_dist];
    }
    else
        fill(dist, own);
    double const *b = x->b;
    double const e = (double)p->epsilon, f = 1.0 - e;
    double best = -1.0;
    codon[0] = codon[1] = codon[2] = 4;
    for (int a = 0; a < 4; ++a)
        for (int bb = 0; bb < 4; ++bb)
            for (int cc = 0; cc < 4; ++cc)
            {
                // the marginal table of "the codon IS (a, bb, cc)": pc where every index is that base or the
                // wildcard, 0 elsewhere -- read entry by entry instead of being built (125 stores per codon,
                // 64 codons per decoded step); the sums see the same operands in the same order, so the
                // arg-max is the one the table gave
                double const pc = x->pc[a * 16 + bb * 4 + cc];
                auto const c3 = [&](int i, int j, int k) {
                    return ((i == 4 || i == a) && (j == 4 || j == bb) && (k == 4 || k == cc)) ? pc : 0.0;
                };
                double const v = frame_prob_word_of(b, c3, e, f, frag, len);
                if (v >= best)
                {
                    best = v;
                    codon[0] = (uint8_t)a, codon[1] = (uint8_t)bb, codon[2] = (uint8_t)cc;
                }
            }
    return best >= 0.0 ? DCP_OK : DCP_EINVAL;
}

char const *dcp_prod_header(void)
{
    return "scan_id\tseq_id\tprofile_name\tabc_name\talt_loglik\tnull_loglik\tprofile_typeid\tversion\tmatch\n";
}

// prod_fwrite + protein_match_write_func (src/server/prod.c:13-41,153-181; protein_match.c:21-56)
long dcp_prod_format_row(char *buf, size_t cap, int64_t scan_id, int64_t seq_id,
                         char const *profile_name, char const *abc_name, double alt_loglik,
                         double null_loglik, char const *profile_typeid, char const *version,
                         dcp_profile const *prof, uint8_t const *seq, unsigned seq_len,
                         struct dcp_step const *steps, unsigned nsteps)
{
    if (!buf || !prof || !seq || (nsteps && !steps)) return -1;
    std::string out;
    char head[512];
    int n = snprintf(head, sizeof head, "%" PRId64 "\t%" PRId64 "\t%s\t%s\t%.17g\t%.17g\t%s\t%s\t", scan_id,
                     seq_id, profile_name ? profile_name : "", abc_name ? abc_name : "", alt_loglik,
                     null_loglik, profile_typeid ? profile_typeid : "", version ? version : "");
    if (n < 0 || (size_t)n >= sizeof head) return -1;
    out.assign(head, (size_t)n);
    unsigned start = 0;
    for (unsigned i = 0; i < nsteps; ++i)
    {
        unsigned const len = steps[i].seqlen, id = steps[i].state_id;
        if (start + len > seq_len) return -1;
        if (i > 0) out.push_back(';');
        for (unsigned k = 0; k < len; ++k)
            out.push_back("ACGT"[seq[start + k] & 3]);
        out.push_back(',');
        char name[8];
        dcp_state_name(id, name);
        out += name;
        out.push_back(',');
        bool const mute = len == 0; // S, B, E, T and D states (protein_state_is_mute)
        if (!mute)
        {
            uint8_t codon[3];
            if (dcp_profile_decode(prof, seq + start, len, id, codon)) return -1;
            for (int k = 0; k < 3; ++k)
                out.push_back("ACGT"[codon[k] & 3]);
            out.push_back(',');
            out.push_back(dcp_gc_decode(codon));
        }
        else
            out.push_back(',');
        start += len;
    }
    out.push_back('\n');
    if (out.size() + 1 > cap) return -1;
    std::memcpy(buf, out.data(), out.size());
    buf[out.size()] = '\0';
    return (long)out.size();
}

} // extern "C"

// ---------------------------------------------------------------------------
// HMMER3 ASCII reader
// ---------------------------------------------------------------------------
#include <sstream>

struct dcp_h3reader
{
    FILE *fp = nullptr;
    bool owns_fp = false;
    int entry_dist;
    float epsilon;
    std::string err;
    unsigned line_no = 0;
    // parameters of the profile the last next() parsed (dcp_h3reader_next_params)
    std::vector<float> trans, match;
    std::string cons, name, acc;
    ~dcp_h3reader()
    {
        if (fp && owns_fp) std::fclose(fp);
    }
};

namespace
{
// one numeric field of a HMMER3 model line: -ln(p), or '*' for p = 0  ->  ln(p)
bool h3_lprob(std::string const &tok, float *out)
{
    if (tok == "*")
    {
        *out = kNegInfF;
        return true;
    }
    char *end = nullptr;
    double v = std::strtod(tok.c_str(), &end);
    if (end == tok.c_str() || *end != '\0' || !(v >= 0.0)) return false;
    *out = (float)(-v);
    if (*out == 0.0f) *out = 0.0f; // no negative zero
    return true;
}

std::vector<std::string> h3_split(std::string const &line)
{
    std::vector<std::string> out;
    std::istringstream ss(line);
    std::string tok;
    while (ss >> tok)
        out.push_back(tok);
    return out;
}
} // namespace

extern "C" {

void dcp_swissprot_null_lprobs(float out[DCP_AMINO_SIZE])
{
    // HMMER3's Swiss-Prot 50.8 amino-acid frequencies, alphabet order ACDEFGHIKLMNPQRSTVWY
    static double const freq[20] = {0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198,
                                    0.0590092, 0.0594422, 0.0963728, 0.0237718, 0.0414386, 0.0482904, 0.0395639,
                                    0.0540978, 0.0683364, 0.0540687, 0.0673417, 0.0114135, 0.0304133};
    for (int i = 0; i < 20; ++i)
        out[i] = (float)std::log(freq[i]);
}

dcp_h3reader *dcp_h3reader_open_fp(FILE *fp, int entry_dist, float epsilon)
{
    if (!fp) return nullptr;
    dcp_h3reader *r = new (std::nothrow) dcp_h3reader();
    if (!r) return nullptr;
    r->fp = fp;
    r->entry_dist = entry_dist;
    r->epsilon = epsilon;
    return r;
}

dcp_h3reader *dcp_h3reader_open(char const *path, int entry_dist, float epsilon)
{
    if (!path) return nullptr;
    FILE *fp = std::fopen(path, "r");
    if (!fp) return nullptr;
    dcp_h3reader *r = dcp_h3reader_open_fp(fp, entry_dist, epsilon);
    if (!r) std::fclose(fp);
    else r->owns_fp = true;
    return r;
}

char const *dcp_h3reader_error(dcp_h3reader const *r) { return r ? r->err.c_str() : "no reader"; }
void dcp_h3reader_close(dcp_h3reader *r) { delete r; }

int dcp_h3reader_next_params(dcp_h3reader *r, struct dcp_h3params *out)
{
    if (!r || !out) return DCP_EINVAL;
    std::memset(out, 0, sizeof *out);
    auto parse_error = [&](std::string const &what) {
        r->err = "line " + std::to_string(r->line_no) + ": " + what;
        return (int)DCP_EPARSE;
    };
    std::string line;
    auto next_line = [&]() -> bool {
        char buf[4096];
        for (;;)
        {
            line.clear();
            bool got = false;
            while (std::fgets(buf, sizeof buf, r->fp)) // a line may be longer than the buffer
            {
                got = true;
                line += buf;
                if (!line.empty() && line.back() == '\n') break;
            }
            if (!got) return false;
            ++r->line_no;
            while (!line.empty() && (line.back() == '\n' || line.back() == '\r'))
                line.pop_back();
            if (line.find_first_not_of(" \t") != std::string::npos) return true;
        }
    };

    // ---- header -----------------------------------------------------------
    if (!next_line()) return DCP_END;
    if (line.compare(0, 6, "HMMER3") != 0) return parse_error("expected a HMMER3 header");
    std::string name, acc, alph;
    unsigned leng = 0;
    bool have_leng = false;
    for (;;)
    {
        if (!next_line()) return parse_error("unexpected end of file in the header");
        std::vector<std::string> f = h3_split(line);
        if (f[0] == "HMM") break;
        if (f.size() >= 2)
        {
            if (f[0] == "NAME") name = f[1];
            else if (f[0] == "ACC") acc = f[1];
            else if (f[0] == "ALPH") alph = f[1];
            else if (f[0] == "LENG")
            {
                char *end = nullptr;
                long v = std::strtol(f[1].c_str(), &end, 10);
                if (*end != '\0' || v < 0) return parse_error("bad LENG");
                leng = (unsigned)v;
                have_leng = true;
            }
        }
    }
    if (!have_leng) return parse_error("missing LENG");
    if (alph != "amino") return parse_error("ALPH must be amino");
    {
        std::vector<std::string> f = h3_split(line); // "HMM A C D ... Y"
        if (f.size() != 21) return parse_error("expected 20 amino-acid columns");
        for (int i = 0; i < 20; ++i)
            if (f[(size_t)i + 1].size() != 1 || f[(size_t)i + 1][0] != kAminoSymbols[i])
                return parse_error("amino-acid columns are not in ACDEFGHIKLMNPQRSTVWY order");
    }
    if (!next_line()) return parse_error("missing the transition header line"); // m->m m->i ...
    if (leng == 0 || leng > DCP_CORE_SIZE_MAX)
    {
        r->err = "core size out of range";
        return DCP_EINVAL; // protein_model_setup: protein_model.c:157-160
    }

    std::vector<float> &trans = r->trans, &match = r->match;
    std::string &cons = r->cons;
    trans.assign((size_t)7 * (leng + 1), 0.0f);
    match.assign((size_t)20 * leng, 0.0f);
    cons.assign(leng, '-');
    auto read_trans = [&](unsigned i) -> bool {
        std::vector<std::string> f = h3_split(line);
        if (f.size() != 7) return false;
        for (int k = 0; k < 7; ++k) // m->m m->i m->d i->m i->i d->m d->d = MM MI MD IM II DM DD
            if (!h3_lprob(f[(size_t)k], &trans[(size_t)7 * i + k])) return false;
        return true;
    };

    // ---- node 0: optional COMPO line, insert emissions, transitions ---------------------------
    if (!next_line()) return parse_error("unexpected end of file");
    if (h3_split(line)[0] == "COMPO" && !next_line()) return parse_error("unexpected end of file");
    if (h3_split(line).size() != 20) return parse_error("expected the insert emissions of node 0");
    if (!next_line() || !read_trans(0)) return parse_error("bad transitions of node 0");

    // ---- nodes 1..LENG -------------------------------------------------------------------------
    for (unsigned k = 1; k <= leng; ++k)
    {
        if (!next_line()) return parse_error("unexpected end of file in the model");
        std::vector<std::string> f = h3_split(line);
        if (f.size() < 21) return parse_error("bad match line");
        char *end = nullptr;
        long idx = std::strtol(f[0].c_str(), &end, 10);
        if (*end != '\0' || idx != (long)k) return parse_error("node index out of sequence");
        for (int a = 0; a < 20; ++a)
            if (!h3_lprob(f[(size_t)a + 1], &match[(size_t)20 * (k - 1) + a])) return parse_error("bad match emission");
        if (f.size() >= 23 && f[22].size() == 1) cons[k - 1] = f[22][0]; // MAP CONS RF MM CS
        if (!next_line() || h3_split(line).size() != 20) return parse_error("bad insert line");
        if (!next_line() || !read_trans(k)) return parse_error("bad transition line");
    }
    if (!next_line() || h3_split(line)[0] != "//") return parse_error("expected the // terminator");

    r->name = name;
    r->acc = acc;
    out->core_size = leng;
    out->match_lprobs = match.data();
    out->trans = trans.data();
    out->consensus = cons.c_str();
    out->name = r->name.c_str();
    out->acc = r->acc.c_str();
    return DCP_OK;
}

int dcp_h3reader_next(dcp_h3reader *r, dcp_profile **out)
{
    if (!r || !out) return DCP_EINVAL;
    *out = nullptr;
    struct dcp_h3params prm;
    int rc = dcp_h3reader_next_params(r, &prm);
    if (rc) return rc;
    float null_lp[20];
    dcp_swissprot_null_lprobs(null_lp);
    char const *label = prm.acc[0] ? prm.acc : prm.name; // hmm.c:113-117
    *out = dcp_profile_new(label, prm.core_size, r->entry_dist, r->epsilon, null_lp, prm.match_lprobs, prm.trans,
                           prm.consensus, &rc);
    if (!*out) r->err = "profile rejected";
    return rc;
}

} // extern "C"

// ---------------------------------------------------------------------------
// dcpx container
// ---------------------------------------------------------------------------
struct dcp_db
{
    FILE *fp = nullptr;
    int entry_dist = 0;
    float epsilon = 0;
    std::vector<uint32_t> sizes;
    int64_t profiles_offset = 0; // where the first profile starts
    std::vector<int64_t> offsets; // per profile
};

namespace
{
constexpr uint16_t kDbMagic = 0xC6F0; // MAGIC_NUMBER, include/deciphon/db/types.h:11
constexpr uint32_t kProfileProtein = 2; // PROFILE_PROTEIN, profile_typeid.h

uint32_t profile_bytes(unsigned M)
{
    unsigned cons = (M + 3u) & ~3u;
    return (uint32_t)(DCP_PROFILE_ACC_SIZE + 4 + cons + 4 * (8 * M + 2 * DCP_NDIST + (size_t)M * DCP_NDIST));
}

template <class T> bool wr(FILE *fp, T const *p, size_t n) { return std::fwrite(p, sizeof(T), n, fp) == n; }
template <class T> bool rd(FILE *fp, T *p, size_t n) { return std::fread(p, sizeof(T), n, fp) == n; }
} // namespace

extern "C" {

int dcp_db_write(char const *path, dcp_profile *const *profiles, unsigned n)
{
    if (!path || !profiles || n == 0 || n > (1u << 20)) return DCP_EINVAL;
    for (unsigned i = 0; i < n; ++i)
        if (!profiles[i] || profiles[i]->entry_dist != profiles[0]->entry_dist ||
            profiles[i]->epsilon != profiles[0]->epsilon)
            return DCP_EINVAL; // one protein_cfg per DB
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return DCP_EIO;
    bool ok = true;
    char const magic[4] = {'D', 'C', 'P', 'X'};
    uint16_t const num = kDbMagic, version = 1;
    uint32_t const typeid_ = kProfileProtein, float_size = 4, entry = (uint32_t)profiles[0]->entry_dist, count = n;
    float const eps = profiles[0]->epsilon;
    char abc[8] = "ACGT", amino[24] = "ACDEFGHIKLMNPQRSTVWY";
    ok = ok && wr(fp, magic, 4) && wr(fp, &num, 1) && wr(fp, &version, 1) && wr(fp, &typeid_, 1) &&
         wr(fp, &float_size, 1) && wr(fp, &entry, 1) && wr(fp, &eps, 1) && wr(fp, abc, 8) && wr(fp, amino, 24) &&
         wr(fp, &count, 1);
    std::vector<uint32_t> sizes(n);
    for (unsigned i = 0; i < n; ++i)
        sizes[i] = profile_bytes(profiles[i]->core_size);
    ok = ok && wr(fp, sizes.data(), n);
    for (unsigned i = 0; ok && i < n; ++i)
    {
        dcp_profile const *p = profiles[i];
        uint32_t const M = p->core_size;
        std::vector<char> cons((M + 3u) & ~3u, '\0');
        std::memcpy(cons.data(), p->consensus.data(), M);
        ok = wr(fp, p->accession, DCP_PROFILE_ACC_SIZE) && wr(fp, &M, 1) && wr(fp, cons.data(), cons.size()) &&
             wr(fp, p->trans8.data(), (size_t)8 * M) && wr(fp, p->null_dist, DCP_NDIST) &&
             wr(fp, p->insert_dist, DCP_NDIST) && wr(fp, p->match_dist.data(), (size_t)M * DCP_NDIST);
    }
    ok = std::fclose(fp) == 0 && ok;
    return ok ? DCP_OK : DCP_EIO;
}

dcp_db *dcp_db_open(char const *path, int *rc)
{
    auto fail = [&](dcp_db *db, int code) -> dcp_db * {
        if (rc) *rc = code;
        if (db)
        {
            if (db->fp) std::fclose(db->fp);
            delete db;
        }
        return nullptr;
    };
    if (!path) return fail(nullptr, DCP_EINVAL);
    dcp_db *db = new (std::nothrow) dcp_db();
    if (!db) return fail(nullptr, DCP_ENOMEM);
    db->fp = std::fopen(path, "rb");
    if (!db->fp) return fail(db, DCP_EIO);
    char magic[4], abc[8], amino[24];
    uint16_t num = 0, version = 0;
    uint32_t typeid_ = 0, float_size = 0, entry = 0, count = 0;
    float eps = 0;
    if (!(rd(db->fp, magic, 4) && rd(db->fp, &num, 1) && rd(db->fp, &version, 1) && rd(db->fp, &typeid_, 1) &&
          rd(db->fp, &float_size, 1) && rd(db->fp, &entry, 1) && rd(db->fp, &eps, 1) && rd(db->fp, abc, 8) &&
          rd(db->fp, amino, 24) && rd(db->fp, &count, 1)))
        return fail(db, DCP_EIO);
    if (std::memcmp(magic, "DCPX", 4) != 0 || num != kDbMagic || version != 1) return fail(db, DCP_EINVAL); // invalid magic number
    if (typeid_ != kProfileProtein) return fail(db, DCP_EINVAL);                                          // invalid typeid
    if (float_size != 4) return fail(db, DCP_EINVAL);                                                     // invalid float size
    if (entry <= DCP_ENTRY_DIST_NULL || entry > DCP_ENTRY_DIST_OCCUPANCY) return fail(db, DCP_EINVAL);   // invalid entry dist
    if (!(eps >= 0.0f && eps <= 1.0f)) return fail(db, DCP_EINVAL);                                       // invalid epsilon
    if (std::memcmp(abc, "ACGT", 5) != 0 || std::memcmp(amino, "ACDEFGHIKLMNPQRSTVWY", 21) != 0) return fail(db, DCP_EINVAL);
    if (count == 0 || count > (1u << 20)) return fail(db, DCP_EINVAL); // MAX_NPROFILES
    db->entry_dist = (int)entry;
    db->epsilon = eps;
    db->sizes.resize(count);
    if (!rd(db->fp, db->sizes.data(), count)) return fail(db, DCP_EIO);
    db->profiles_offset = (int64_t)std::ftell(db->fp);
    db->offsets.resize((size_t)count + 1);
    db->offsets[0] = db->profiles_offset;
    for (uint32_t i = 0; i < count; ++i)
        db->offsets[i + 1] = db->offsets[i] + db->sizes[i];
    if (std::fseek(db->fp, 0, SEEK_END) != 0 || (int64_t)std::ftell(db->fp) != db->offsets[count])
        return fail(db, DCP_EIO); // truncated or trailing bytes
    if (rc) *rc = DCP_OK;
    return db;
}

void dcp_db_close(dcp_db *db)
{
    if (!db) return;
    if (db->fp) std::fclose(db->fp);
    delete db;
}

unsigned dcp_db_nprofiles(dcp_db const *db) { return (unsigned)db->sizes.size(); }
int dcp_db_entry_dist(dcp_db const *db) { return db->entry_dist; }
float dcp_db_epsilon(dcp_db const *db) { return db->epsilon; }
uint32_t const *dcp_db_profile_sizes(dcp_db const *db) { return db->sizes.data(); }

unsigned dcp_db_partitions(dcp_db const *db, unsigned npartitions, unsigned part_size[DCP_NUM_THREADS],
                           int64_t part_offset[DCP_NUM_THREADS + 1])
{
    // partition_init + partition_it, src/db/profile_reader.c:45-72
    unsigned const nprofs = dcp_db_nprofiles(db);
    unsigned const nparts = dcp_partition_by_count(nprofs, npartitions, part_size);
    if (nparts == 0) return 0;
    for (unsigned i = 0; i <= DCP_NUM_THREADS; ++i)
        part_offset[i] = 0;
    part_offset[0] = db->profiles_offset;
    unsigned i = 0, size = 0;
    for (unsigned j = 0; j < nprofs; ++j)
    {
        part_offset[i + 1] += db->sizes[j];
        if (++size >= part_size[i])
        {
            part_offset[i + 1] += part_offset[i];
            ++i;
            size = 0;
        }
    }
    return nparts;
}

int dcp_db_read(dcp_db *db, unsigned begin, unsigned end, dcp_profile **out)
{
    if (!db || !out || begin > end || end > dcp_db_nprofiles(db)) return DCP_EINVAL;
    if (std::fseek(db->fp, (long)db->offsets[begin], SEEK_SET) != 0) return DCP_EIO;
    for (unsigned i = begin; i < end; ++i)
        out[i - begin] = nullptr;
    for (unsigned i = begin; i < end; ++i)
    {
        dcp_profile *p = new (std::nothrow) dcp_profile();
        if (!p) return DCP_ENOMEM;
        out[i - begin] = p;
        uint32_t M = 0;
        if (!(rd(db->fp, p->accession, DCP_PROFILE_ACC_SIZE) && rd(db->fp, &M, 1))) return DCP_EIO;
        p->accession[DCP_PROFILE_ACC_SIZE - 1] = '\0';
        if (M == 0 || M > DCP_CORE_SIZE_MAX || profile_bytes(M) != db->sizes[i]) return DCP_EPARSE; // profile is too long
        std::vector<char> cons((M + 3u) & ~3u);
        p->core_size = M;
        p->entry_dist = db->entry_dist;
        p->epsilon = db->epsilon;
        p->trans8.resize((size_t)8 * M);
        p->match_dist.resize((size_t)M * DCP_NDIST);
        if (!(rd(db->fp, cons.data(), cons.size()) && rd(db->fp, p->trans8.data(), (size_t)8 * M) &&
              rd(db->fp, p->null_dist, DCP_NDIST) && rd(db->fp, p->insert_dist, DCP_NDIST) &&
              rd(db->fp, p->match_dist.data(), (size_t)M * DCP_NDIST)))
            return DCP_EIO;
        p->consensus.assign(cons.begin(), cons.begin() + M);
        p->consensus.push_back('\0');
    }
    return DCP_OK;
}

} // extern "C"

