/*
 * Mutation fuzzer for the .dcp database path of the host layer (MessagePack framing, header checks,
 * profile map(16), nuclt_dists, dp values, partition table).  Built from the host layer's C files with
 * -fsanitize=address,undefined by tests/test_sanitizers.py: every mutated file must end in RC_OK /
 * RC_END or a clean error code -- never a crash, an out-of-bounds access, a leak or a hang.
 *   fuzz_dcp scratch_file iterations seed
 */
#include "deciphon_host.h"
#include <stdlib.h>
#include <string.h>

static uint64_t rng_state;
static uint64_t rnd(void)
{
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return rng_state;
}

static unsigned char *good;
static size_t good_len;

static void press(char const *path)
{
    struct imm_nuclt const *nuclt = imm_super(&imm_dna_iupac);
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, nuclt);
    FILE *fp = fopen(path, "wb+");
    if (!fp) exit(2);
    struct protein_db_writer w = {0};
    if (protein_db_writer_open(&w, fp, &imm_amino_iupac, nuclt, PROTEIN_CFG_DEFAULT)) exit(2);
    unsigned const sizes[4] = {2, 9, 31, 3};
    for (unsigned p = 0; p < 4; ++p)
    {
        struct protein_profile prof;
        char acc[16];
        snprintf(acc, sizeof acc, "PF%05u", p);
        protein_profile_init(&prof, acc, &imm_amino_iupac, &code, PROTEIN_CFG_DEFAULT);
        if (protein_profile_sample(&prof, 7 + p, sizes[p]) || protein_db_writer_pack_profile(&w, &prof)) exit(2);
        profile_del(&prof.super);
    }
    if (db_writer_close((struct db_writer *)&w, true)) exit(2);
    good_len = (size_t)ftell(fp);
    good = malloc(good_len);
    rewind(fp);
    if (fread(good, 1, good_len, fp) != good_len) exit(2);
    fclose(fp);
}

static unsigned long outcomes[9];

static void read_all(char const *path, unsigned nparts, int balanced)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) exit(2);
    struct protein_db_reader db = {0};
    enum rc rc = protein_db_reader_open(&db, fp);
    if (!rc)
    {
        static struct profile_reader reader;
        rc = balanced ? profile_reader_setup_balanced(&reader, (struct db_reader *)&db, nparts)
                      : profile_reader_setup(&reader, (struct db_reader *)&db, nparts);
        if (!rc)
        {
            for (unsigned i = 0; i < profile_reader_npartitions(&reader); ++i)
            {
                struct profile *prof = NULL;
                unsigned n = 0;
                while ((rc = profile_reader_next(&reader, i, &prof)) == RC_OK)
                {
                    struct protein_profile *pp = (struct protein_profile *)prof;
                    /* touch what a scan would read */
                    if (pp->core_size == 0 || pp->core_size > PROTEIN_MODEL_CORE_SIZE_MAX || !pp->impl) abort();
                    if (dcp_profile_core_size(pp->impl) != pp->core_size) abort();
                    volatile float sink = dcp_profile_trans8(pp->impl)[8 * pp->core_size - 1] +
                                          dcp_profile_match_dist(pp->impl)[DCP_NDIST * pp->core_size - 1];
                    (void)sink;
                    if (++n > MAX_NPROFILES) abort(); /* a reader that never ends */
                }
            }
            profile_reader_del(&reader);
        }
        db_reader_close((struct db_reader *)&db);
    }
    outcomes[rc < 9 ? rc : 8]++;
    fclose(fp);
}

int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    char const *path = argv[1];
    unsigned long const iters = strtoul(argv[2], NULL, 10);
    rng_state = strtoull(argv[3], NULL, 10) | 1;
    if (!freopen("/dev/null", "w", stderr)) return 2; /* the host layer logs every error it returns */
    press(path);
    read_all(path, 3, 0);
    if (outcomes[RC_END] != 1) return 3; /* the untouched file reads to its end */
    unsigned char *buf = malloc(good_len * 2 + 64);
    for (unsigned long it = 0; it < iters; ++it)
    {
        size_t len = good_len;
        memcpy(buf, good, len);
        unsigned const nmut = 1 + (unsigned)(rnd() % 4);
        for (unsigned m = 0; m < nmut && len; ++m)
        {
            size_t const at = rnd() % len;
            switch (rnd() % 6)
            {
            case 0: buf[at] = (unsigned char)rnd(); break;
            case 1: len = at; break; /* truncate */
            case 2:
            { /* duplicate a run */
                size_t const n = 1 + rnd() % 64;
                size_t const k = at + n <= len ? n : len - at;
                if (len + k <= good_len * 2)
                {
                    memmove(buf + at + k, buf + at, len - at);
                    len += k;
                }
                break;
            }
            case 3:
            { /* delete a run */
                size_t const n = 1 + rnd() % 64;
                size_t const k = at + n <= len ? n : len - at;
                memmove(buf + at, buf + at + k, len - at - k);
                len -= k;
                break;
            }
            case 4: /* a length / count field blown up */
                buf[at] = (unsigned char)(0xc4 + rnd() % 28);
                break;
            default: buf[at] ^= (unsigned char)(1u << (rnd() % 8)); break;
            }
        }
        FILE *fp = fopen(path, "wb");
        if (!fp) return 2;
        if (len) fwrite(buf, 1, len, fp);
        fclose(fp);
        read_all(path, 1 + (unsigned)(rnd() % 5), (int)(rnd() & 1));
    }
    free(buf);
    free(good);
    remove(path);
    printf("fuzz_dcp ok: %lu iterations; outcomes OK/END %lu/%lu EFAIL %lu EINVAL %lu EIO %lu ENOMEM %lu EPARSE %lu\n", iters,
           outcomes[RC_OK], outcomes[RC_END], outcomes[RC_EFAIL], outcomes[RC_EINVAL], outcomes[RC_EIO],
           outcomes[RC_ENOMEM], outcomes[RC_EPARSE]);
    return 0;
}
