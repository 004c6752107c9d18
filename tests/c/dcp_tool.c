/*
 * Test helper (tests/test_oracle_io.py): drives the PRODUCT's host layer by the reference's own entry
 * points and leaves plain binary files for the Python side to compare with the oracle's independent
 * readers (oracle/oracle_io.c) of the same bytes.
 *
 *   dcp_tool press <out.dcp> <sidecar.bin> <n> <seed0>     (CPU)
 *       protein_profile_sample x n -> protein_db_writer_* -> out.dcp, the way hmm_press writes a database
 *       (src/server/hmm.c:120-178, test/protein_db.c:18-50).  sidecar: per profile u32 M, then the
 *       in-memory values float trans8[8][M], null[129], insert[129], match[M][129].
 *   dcp_tool scan <in.dcp> <seqs.txt> <scores.bin> <multi_hits> <hmmer3_compat>     (GPU)
 *       protein_db_reader_open + profile_reader_* -> for every (profile, sequence) what thread_run does
 *       per pair (src/server/scan_thread.c:99-117): protein_profile_setup, imm_task_setup,
 *       imm_dp_viterbi(null), imm_dp_viterbi(alt).  scores: float [nprofiles][nseqs][2] = {null, alt}.
 *   dcp_tool hmm <in.hmm> <seqs.txt> <scores.bin> <multi_hits> <hmmer3_compat> <entry_dist> <epsilon>   (GPU)
 *       protein_h3reader_init / _next + protein_profile_absorb (hmm_press's reader) -> the same per-pair calls.
 */
#include "deciphon_host.h"
#include <stdlib.h>
#include <string.h>

#define DIE(...)                                                                                                   \
    do                                                                                                             \
    {                                                                                                              \
        fprintf(stderr, "dcp_tool: " __VA_ARGS__);                                                                 \
        fprintf(stderr, "\n");                                                                                     \
        exit(2);                                                                                                   \
    } while (0)

static char **read_seqs(char const *path, unsigned *n)
{
    FILE *fp = fopen(path, "r");
    if (!fp) DIE("cannot open %s", path);
    char **out = NULL;
    *n = 0;
    static char line[1 << 21];
    while (fgets(line, sizeof line, fp))
    {
        size_t len = strcspn(line, "\r\n");
        line[len] = 0;
        if (!len) continue;
        out = realloc(out, sizeof *out * (*n + 1));
        out[(*n)++] = strdup(line);
    }
    fclose(fp);
    return out;
}

static void put(FILE *fp, void const *p, size_t bytes)
{
    if (fwrite(p, 1, bytes, fp) != bytes) DIE("short write");
}

/* one pair the way thread_run scores it; the two tasks live as long as the profile's dps do */
static void score_pair(struct protein_profile *prof, struct imm_task **tn, struct imm_task **ta, char const *text,
                       bool multi_hits, bool hmmer3_compat, float out[2])
{
    struct imm_seq seq = imm_seq(imm_str(text), prof->super.code->abc);
    if (protein_profile_setup(prof, imm_seq_size(&seq), multi_hits, hmmer3_compat)) DIE("protein_profile_setup");
    struct imm_prod prod = imm_prod();
    if (!*tn) *tn = imm_task_new(&prof->null.dp);
    else if (imm_task_reset(*tn, &prof->null.dp)) DIE("imm_task_reset");
    if (!*ta) *ta = imm_task_new(&prof->alt.dp);
    else if (imm_task_reset(*ta, &prof->alt.dp)) DIE("imm_task_reset");
    if (!*tn || !*ta) DIE("imm_task_new");
    if (imm_task_setup(*tn, &seq) || imm_dp_viterbi(&prof->null.dp, *tn, &prod)) DIE("null viterbi");
    out[0] = prod.loglik;
    imm_prod_reset(&prod);
    if (imm_task_setup(*ta, &seq) || imm_dp_viterbi(&prof->alt.dp, *ta, &prod)) DIE("alt viterbi");
    out[1] = prod.loglik;
    imm_del(&prod);
}

static int cmd_press(char **a)
{
    unsigned const n = (unsigned)atoi(a[2]), seed0 = (unsigned)atoi(a[3]);
    struct imm_nuclt const *nuclt = imm_super(&imm_dna_iupac);
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, nuclt);
    FILE *fp = fopen(a[0], "wb"), *side = fopen(a[1], "wb");
    if (!fp || !side) DIE("cannot create outputs");
    struct protein_db_writer db;
    if (protein_db_writer_open(&db, fp, &imm_amino_iupac, nuclt, PROTEIN_CFG_DEFAULT)) DIE("writer open");
    for (unsigned i = 0; i < n; ++i)
    {
        struct protein_profile prof;
        char acc[32];
        snprintf(acc, sizeof acc, "PF%05u.%u", i, i % 7);
        protein_profile_init(&prof, acc, &imm_amino_iupac, &code, PROTEIN_CFG_DEFAULT);
        unsigned const M = 2 + (i * 37) % 140 + (i == n - 1 ? 300 : 0); /* 2 .. 141, one big one */
        if (protein_profile_sample(&prof, seed0 + i, M)) DIE("sample");
        if (protein_db_writer_pack_profile(&db, &prof)) DIE("pack");
        uint32_t m32 = M;
        put(side, &m32, 4);
        put(side, dcp_profile_trans8(prof.impl), sizeof(float) * 8 * M);
        put(side, dcp_profile_null_dist(prof.impl), sizeof(float) * DCP_NDIST);
        put(side, dcp_profile_insert_dist(prof.impl), sizeof(float) * DCP_NDIST);
        put(side, dcp_profile_match_dist(prof.impl), sizeof(float) * DCP_NDIST * M);
        profile_del((struct profile *)&prof);
    }
    if (db_writer_close((struct db_writer *)&db, true)) DIE("writer close");
    fclose(fp);
    fclose(side);
    return 0;
}

static int cmd_scan(char **a)
{
    unsigned nseqs = 0;
    char **seqs = read_seqs(a[1], &nseqs);
    bool const multi = atoi(a[3]) != 0, h3 = atoi(a[4]) != 0;
    FILE *fp = fopen(a[0], "rb"), *out = fopen(a[2], "wb");
    if (!fp || !out) DIE("cannot open files");
    struct protein_db_reader db;
    if (protein_db_reader_open(&db, fp)) DIE("reader open");
    struct profile_reader reader;
    if (profile_reader_setup(&reader, (struct db_reader *)&db, 1)) DIE("profile_reader_setup");
    struct profile *prof = NULL;
    enum rc rc;
    unsigned np = 0;
    while ((rc = profile_reader_next(&reader, 0, &prof)) == RC_OK)
    {
        struct imm_task *tn = NULL, *ta = NULL;
        for (unsigned q = 0; q < nseqs; ++q)
        {
            float s[2];
            score_pair((struct protein_profile *)prof, &tn, &ta, seqs[q], multi, h3, s);
            put(out, s, sizeof s);
        }
        imm_del(tn);
        imm_del(ta);
        ++np;
    }
    if (rc != RC_END) DIE("profile_reader_next: rc %d after %u profiles", (int)rc, np);
    profile_reader_del(&reader);
    db_reader_close((struct db_reader *)&db);
    fclose(fp);
    fclose(out);
    return 0;
}

static int cmd_hmm(char **a)
{
    unsigned nseqs = 0;
    char **seqs = read_seqs(a[1], &nseqs);
    bool const multi = atoi(a[3]) != 0, h3 = atoi(a[4]) != 0;
    struct protein_cfg cfg = protein_cfg((enum entry_dist)atoi(a[5]), (imm_float)atof(a[6]));
    FILE *fp = fopen(a[0], "r"), *out = fopen(a[2], "wb");
    if (!fp || !out) DIE("cannot open files");
    struct imm_nuclt const *nuclt = imm_super(&imm_dna_iupac);
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, nuclt);
    struct protein_h3reader reader;
    protein_h3reader_init(&reader, &imm_amino_iupac, &code, cfg, fp);
    enum rc rc;
    while ((rc = protein_h3reader_next(&reader)) == RC_OK)
    {
        struct protein_profile prof;
        protein_profile_init(&prof, reader.acc, &imm_amino_iupac, &code, cfg);
        if (protein_profile_absorb(&prof, &reader.model)) DIE("absorb");
        struct imm_task *tn = NULL, *ta = NULL;
        for (unsigned q = 0; q < nseqs; ++q)
        {
            float s[2];
            score_pair(&prof, &tn, &ta, seqs[q], multi, h3, s);
            put(out, s, sizeof s);
        }
        imm_del(tn);
        imm_del(ta);
        profile_del((struct profile *)&prof);
    }
    if (rc != RC_END) DIE("protein_h3reader_next: rc %d", (int)rc);
    protein_h3reader_del(&reader);
    fclose(fp);
    fclose(out);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 6 && !strcmp(argv[1], "press")) return cmd_press(argv + 2);
    if (argc == 7 && !strcmp(argv[1], "scan")) return cmd_scan(argv + 2);
    if (argc == 9 && !strcmp(argv[1], "hmm")) return cmd_hmm(argv + 2);
    fprintf(stderr, "usage: dcp_tool press|scan|hmm ... (see the header of tests/c/dcp_tool.c)\n");
    return 1;
}
