/* oracle/selftest.c -- sanitizer driver for the CPU oracle (test infrastructure).
 * Built with -fsanitize=address,undefined by `make -C oracle check-asan`: runs the reference's
 * goldens G1-G3 (test/protein_profile.c:41,65,157) and cross-checks the generic graph Viterbi
 * against the end-indexed recursion on random profiles/sequences, incl. path scoring. */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static int failed;
#define CHECK(c)                                                       \
    do                                                                 \
    {                                                                  \
        if (!(c)) { fprintf(stderr, "selftest:%d: %s\n", __LINE__, #c); failed++; } \
    } while (0)

static void encode(char const *s, unsigned char *out, unsigned n)
{
    for (unsigned i = 0; i < n; ++i)
        out[i] = s[i] == 'A' ? 0 : s[i] == 'C' ? 1 : s[i] == 'G' ? 2 : 3;
}

int main(void)
{
    char const q[] = "ATGAAACGCATTAGCACCACCATTACCACCAC";
    unsigned char seq[32];
    encode(q, seq, 32);
    double const tol = sizeof(ofloat) == 8 ? 1e-9 : 5e-5;
    for (int entry = 1; entry <= 2; ++entry)
    {
        struct orc_profile *p = orc_profile_sample(1, 2, entry, (ofloat)0.1f);
        CHECK(p != NULL);
        CHECK(orc_profile_setup(p, 0, 1, 0) == ORC_EINVAL);
        CHECK(orc_profile_setup(p, 32, 1, 0) == ORC_OK);
        ofloat ll;
        uint16_t st[128];
        uint8_t ln[128];
        unsigned n = 128;
        CHECK(orc_viterbi(p, 0, seq, 32, &ll, st, ln, &n) == ORC_OK);
        CHECK(fabs((double)ll - (-48.9272687711)) <= tol * 48.93 && n == 11);
        n = 128;
        CHECK(orc_viterbi(p, 1, seq, 32, &ll, st, ln, &n) == ORC_OK);
        double want = entry == 1 ? -55.59428153448 : -54.35543421312;
        CHECK(fabs((double)ll - want) <= tol * fabs(want) && n == 14);
        CHECK(orc_path_score(p, 1, seq, 32, st, ln, n) == ll);
        unsigned char codon[3];
        unsigned pos = 0, emitted = 0;
        for (unsigned i = 0; i < n; ++i)
            if (ln[i])
            {
                orc_profile_decode(p, seq + pos, ln[i], st[i], codon);
                pos += ln[i];
                emitted++;
            }
        CHECK(pos == 32 && emitted == 10);
        orc_profile_del(p);
    }
    struct orc_rnd rnd;
    orc_rnd_seed(&rnd, 99);
    for (unsigned t = 0; t < 40; ++t)
    {
        unsigned M = 2 + (unsigned)(orc_rnd_dbl(&rnd) * 70), L = 1 + (unsigned)(orc_rnd_dbl(&rnd) * 90);
        struct orc_profile *p = orc_profile_sample(100 + t, M, 1 + (int)(t & 1), (ofloat)0.01f);
        unsigned char *s = malloc(L);
        for (unsigned i = 0; i < L; ++i)
            s[i] = (unsigned char)(orc_rnd_dbl(&rnd) * 4) & 3;
        CHECK(orc_profile_setup(p, L, (int)(t % 3 != 0), (int)(t % 5 == 0)) == ORC_OK);
        ofloat g0, g1, f0, f1;
        unsigned cap = 2 * L + 3 * M + 16, n = cap;
        uint16_t *st = malloc(sizeof *st * cap);
        uint8_t *ln = malloc(cap);
        CHECK(orc_viterbi(p, 0, s, L, &g0, NULL, NULL, &n) == ORC_OK);
        n = cap;
        CHECK(orc_viterbi(p, 1, s, L, &g1, st, ln, &n) == ORC_OK);
        CHECK(orc_viterbi_fast(p, s, L, &f0, &f1) == ORC_OK);
        CHECK(g0 == f0 && g1 == f1); /* same float association: bit-equal */
        CHECK(orc_path_score(p, 1, s, L, st, ln, n) == g1);
        free(st), free(ln), free(s);
        orc_profile_del(p);
    }
    if (!failed) puts("oracle selftest ok");
    return failed;
}
