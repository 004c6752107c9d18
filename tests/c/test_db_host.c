/*
 * CPU-only test of the host layer's database path (no device call): press profiles into a .dcp
 * file (protein_db_writer, as test/protein_db.c:18-50 does), read it back (protein_db_reader,
 * profile_reader) and require every stored value to survive bit for bit; the partition table of
 * profile_reader_setup (src/db/profile_reader.c:45-72); the reference's header checks
 * (src/db/reader.c:25-79, src/db/protein_reader.c:8-29); and the N2 framing reader's behaviour on a
 * profile whose dp values are in a foreign (imm's) serialisation.
 * Exit status = number of failed checks.
 */
#include "deciphon_host.h"
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int failed;
#define CHECK(cond)                                                                        \
    do                                                                                     \
    {                                                                                      \
        if (!(cond))                                                                       \
        {                                                                                  \
            fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond);       \
            failed++;                                                                      \
        }                                                                                  \
    } while (0)

enum { NPROF = 7 };
static unsigned const kSizes[NPROF] = {2, 17, 300, 5, 64, 1, 129};

static FILE *tmp(char path[64])
{
    snprintf(path, 64, "/tmp/dcp_test_db_XXXXXX");
    int fd = mkstemp(path);
    return fd < 0 ? NULL : fdopen(fd, "wb+");
}

static void make_profile(struct protein_profile *prof, struct imm_nuclt_code const *code, unsigned p)
{
    char acc[16];
    snprintf(acc, sizeof acc, "PF%05u.%u", p, p + 1);
    protein_profile_init(prof, acc, &imm_amino_iupac, code, PROTEIN_CFG_DEFAULT);
    if (kSizes[p] >= 2) CHECK(protein_profile_sample(prof, 100 + p, kSizes[p]) == RC_OK);
    else
    {
        /* protein_profile_sample asserts core_size >= 2: a 1-node profile through the model builder */
        struct imm_rnd rnd = imm_rnd(5);
        imm_float null[IMM_AMINO_SIZE], match[IMM_AMINO_SIZE];
        struct protein_trans t[2];
        imm_lprob_sample(&rnd, IMM_AMINO_SIZE, null);
        imm_lprob_normalize(IMM_AMINO_SIZE, null);
        imm_lprob_sample(&rnd, IMM_AMINO_SIZE, match);
        imm_lprob_normalize(IMM_AMINO_SIZE, match);
        for (int i = 0; i < 2; ++i)
        {
            imm_lprob_sample(&rnd, PROTEIN_TRANS_SIZE, t[i].data);
            imm_lprob_normalize(PROTEIN_TRANS_SIZE, t[i].data);
        }
        struct protein_model model;
        protein_model_init(&model, &imm_amino_iupac, code, PROTEIN_CFG_DEFAULT, null);
        CHECK(protein_model_add_node(&model, match, 'K') == RC_EFAIL); /* setup first */
        CHECK(protein_model_setup(&model, 0) == RC_EINVAL);
        CHECK(protein_model_setup(&model, PROTEIN_MODEL_CORE_SIZE_MAX + 1) == RC_EINVAL);
        CHECK(protein_model_setup(&model, 1) == RC_OK);
        CHECK(protein_profile_absorb(prof, &model) == RC_EINVAL); /* incomplete */
        CHECK(protein_model_add_node(&model, match, 'K') == RC_OK);
        CHECK(protein_model_add_node(&model, match, 'K') == RC_EFAIL); /* limit of nodes */
        CHECK(protein_model_add_trans(&model, t[0]) == RC_OK);
        CHECK(protein_model_add_trans(&model, t[1]) == RC_OK);
        CHECK(protein_model_add_trans(&model, t[1]) == RC_EFAIL); /* limit of transitions */
        CHECK(protein_profile_absorb(prof, &model) == RC_OK);
        CHECK(prof->core_size == 1 && prof->consensus[0] == 'K');
        protein_model_del(&model);
    }
}

static int same_profile(struct protein_profile const *a, struct protein_profile const *b)
{
    unsigned const M = a->core_size;
    return M == b->core_size && !strcmp(a->super.accession, b->super.accession) &&
           !strcmp(a->consensus, b->consensus) &&
           !memcmp(dcp_profile_trans8(a->impl), dcp_profile_trans8(b->impl), sizeof(float) * 8 * M) &&
           !memcmp(dcp_profile_match_dist(a->impl), dcp_profile_match_dist(b->impl), sizeof(float) * DCP_NDIST * M) &&
           !memcmp(dcp_profile_null_dist(a->impl), dcp_profile_null_dist(b->impl), sizeof(float) * DCP_NDIST) &&
           !memcmp(dcp_profile_insert_dist(a->impl), dcp_profile_insert_dist(b->impl), sizeof(float) * DCP_NDIST) &&
           !memcmp(a->alt.match_ndists[M - 1].codonm.lprobs, b->alt.match_ndists[M - 1].codonm.lprobs,
                   sizeof a->alt.match_ndists[0].codonm.lprobs) &&
           !memcmp(a->null.ndist.nucltp.lprobs, b->null.ndist.nucltp.lprobs, sizeof a->null.ndist.nucltp.lprobs) &&
           !memcmp(a->xtrans, b->xtrans, sizeof a->xtrans) && a->alt.T == b->alt.T && a->null.R == b->null.R;
}

static void roundtrip(void)
{
    struct imm_nuclt const *nuclt = imm_super(&imm_dna_iupac);
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, nuclt);
    char path[64];
    FILE *fp = tmp(path);
    CHECK(fp != NULL);
    struct protein_db_writer w = {0};
    CHECK(protein_db_writer_open(&w, fp, &imm_amino_iupac, nuclt, PROTEIN_CFG_DEFAULT) == RC_OK);
    static struct protein_profile src[NPROF];
    for (unsigned p = 0; p < NPROF; ++p)
    {
        make_profile(&src[p], &code, p);
        CHECK(protein_db_writer_pack_profile(&w, &src[p]) == RC_OK);
    }
    CHECK(db_writer_close((struct db_writer *)&w, true) == RC_OK);
    long const file_size = ftell(fp);
    CHECK(file_size > 0);
    rewind(fp);

    struct protein_db_reader db = {0};
    CHECK(protein_db_reader_open(&db, fp) == RC_OK);
    CHECK(db.super.nprofiles == NPROF && db.super.profile_typeid == PROFILE_PROTEIN);
    CHECK(!strcmp(imm_abc_symbols(imm_super(&db.nuclt)), "ACGT") && imm_abc_typeid(imm_super(&db.nuclt)) == IMM_DNA);
    CHECK(!strcmp(imm_abc_symbols(imm_super(&db.amino)), "ACDEFGHIKLMNPQRSTVWY"));
    CHECK(imm_abc_any_symbol_id(imm_super(&db.nuclt)) == 4 && db.code.nuclt == &db.nuclt);
    CHECK(db.cfg.entry_dist == ENTRY_DIST_OCCUPANCY && db.cfg.epsilon == DEFAULT_EPSILON);

    /* partition table == partition_it (profile_reader.c:54-72), incl. the unwritten end offset of trailing
     * empty partitions: 7 profiles over 5 partitions -> ceil sizes 2,2,2,1 and an empty fifth */
    static struct profile_reader reader;
    CHECK(profile_reader_setup(&reader, (struct db_reader *)&db, 5) == RC_OK);
    CHECK(profile_reader_npartitions(&reader) == 5 && profile_reader_nprofiles(&reader) == NPROF);
    unsigned const want_sizes[5] = {2, 2, 2, 1, 0};
    int64_t off = reader.partition_offset[0];
    uint64_t sum_sizes = 0;
    unsigned j = 0;
    for (unsigned i = 0; i < 5; ++i)
    {
        CHECK(profile_reader_partition_size(&reader, i) == want_sizes[i]);
        for (unsigned k = 0; k < want_sizes[i]; ++k, ++j)
            off += db.super.profile_sizes[j];
        if (want_sizes[i]) CHECK(reader.partition_offset[i + 1] == off);
    }
    CHECK(reader.partition_offset[5] == 0); /* the reference's quirk */
    for (j = 0; j < NPROF; ++j)
        sum_sizes += db.super.profile_sizes[j];
    CHECK(reader.partition_offset[0] + (int64_t)sum_sizes == file_size); /* the profiles end the file */

    /* every profile back, bit for bit, through next() of its partition */
    j = 0;
    for (unsigned i = 0; i < 5; ++i)
    {
        struct profile *prof = NULL;
        enum rc rc;
        while ((rc = profile_reader_next(&reader, i, &prof)) == RC_OK)
        {
            CHECK(j < NPROF && same_profile((struct protein_profile *)prof, &src[j]));
            ++j;
        }
        CHECK(rc == RC_END);
    }
    CHECK(j == NPROF);
    /* rewind + second pass of one partition */
    CHECK(profile_reader_rewind(&reader, 2) == RC_OK);
    struct profile *prof = NULL;
    CHECK(profile_reader_next(&reader, 2, &prof) == RC_OK);
    CHECK(same_profile((struct protein_profile *)prof, &src[4]));
    /* setup changes the specials of THIS object only, through imm_dp_trans_idx / imm_dp_change_trans */
    struct protein_profile *pp = (struct protein_profile *)prof;
    CHECK(pp->xtrans[0] == 0.0f && pp->xtrans[9] == 0.0f);
    CHECK(protein_profile_setup(pp, 0, true, false) == RC_EINVAL);
    CHECK(protein_profile_setup(pp, 100, true, false) == RC_OK);
    float xt[DCP_NXTRANS];
    CHECK(dcp_xtrans(100, 1, 0, xt) == 0 && !memcmp(xt, pp->xtrans, sizeof xt));
    CHECK(imm_dp_trans_idx(&pp->alt.dp, pp->alt.E, pp->alt.B) == 9);
    CHECK(imm_dp_trans_idx(&pp->alt.dp, pp->alt.B, pp->alt.E) == UINT_MAX);
    CHECK(imm_dp_trans_idx(&pp->null.dp, pp->null.R, pp->null.R) == 0);
    profile_reader_del(&reader);

    /* balanced partitions: same invariants, boundaries by bytes (~ cells) instead of by count */
    rewind(fp);
    db_reader_close((struct db_reader *)&db);
    CHECK(protein_db_reader_open(&db, fp) == RC_OK);
    CHECK(profile_reader_setup_balanced(&reader, (struct db_reader *)&db, 3) == RC_OK);
    CHECK(profile_reader_npartitions(&reader) == 3 && profile_reader_nprofiles(&reader) == NPROF);
    uint64_t worst = 0;
    j = 0;
    for (unsigned i = 0; i < 3; ++i)
    {
        uint64_t bytes = 0;
        CHECK(profile_reader_partition_size(&reader, i) >= 1);
        for (unsigned k = 0; k < profile_reader_partition_size(&reader, i); ++k, ++j)
            bytes += db.super.profile_sizes[j];
        CHECK(reader.partition_offset[i + 1] - reader.partition_offset[i] == (int64_t)bytes);
        if (bytes > worst) worst = bytes;
    }
    /* the 300-node profile alone is more than a third: the best any contiguous split can do is put it
     * with as little else as possible -- {2,17} {300} {5,64,1,129} or similar, never 300 + 64 + 129 */
    CHECK(worst < sum_sizes * 2 / 3);
    profile_reader_del(&reader);
    db_reader_close((struct db_reader *)&db);

    /* truncation anywhere in the profiles is an error of the read, never a short profile */
    for (long cut = file_size - 1; cut > file_size - 4000; cut -= 997)
    {
        CHECK(ftruncate(fileno(fp), cut) == 0);
        rewind(fp);
        struct protein_db_reader t = {0};
        CHECK(protein_db_reader_open(&t, fp) == RC_OK);
        CHECK(profile_reader_setup(&reader, (struct db_reader *)&t, 1) == RC_OK);
        enum rc rc;
        unsigned n = 0;
        while ((rc = profile_reader_next(&reader, 0, &prof)) == RC_OK)
            ++n;
        CHECK(rc != RC_END && n == NPROF - 1);
        profile_reader_del(&reader);
        db_reader_close((struct db_reader *)&t);
    }
    for (unsigned p = 0; p < NPROF; ++p)
        profile_del(&src[p].super);
    fclose(fp);
    remove(path);
}

/* header checks of src/db/reader.c + protein_reader.c: each wrong field is RC_EINVAL, a missing key RC_EIO */
static enum rc open_header(unsigned magic, unsigned typeid, unsigned fsize, unsigned edist, float eps, int drop_key)
{
    char path[64];
    FILE *fp = tmp(path);
    struct lip_file f;
    lip_file_init(&f, fp);
    lip_write_map_size(&f, 2);
    lip_write_cstr(&f, "header");
    lip_write_map_size(&f, 8);
    lip_write_cstr(&f, drop_key == 0 ? "magic" : "magic_number"), lip_write_int(&f, magic);
    lip_write_cstr(&f, "profile_typeid"), lip_write_int(&f, typeid);
    lip_write_cstr(&f, "float_size"), lip_write_int(&f, fsize);
    lip_write_cstr(&f, "entry_dist"), lip_write_int(&f, edist);
    lip_write_cstr(&f, "epsilon"), lip_write_float(&f, eps);
    lip_write_cstr(&f, "abc"), imm_abc_pack(imm_super(imm_super(&imm_dna_iupac)), &f);
    lip_write_cstr(&f, "amino"), imm_abc_pack(imm_super(&imm_amino_iupac), &f);
    lip_write_cstr(&f, "profile_sizes"), lip_write_1darray_size_type(&f, 0, LIP_1DARRAY_UINT32);
    lip_write_cstr(&f, "profiles"), lip_write_array_size(&f, 0);
    rewind(fp);
    struct protein_db_reader db = {0};
    enum rc rc = protein_db_reader_open(&db, fp);
    if (!rc) db_reader_close((struct db_reader *)&db);
    fclose(fp);
    remove(path);
    return rc;
}

static void header_checks(void)
{
    CHECK(open_header(MAGIC_NUMBER, PROFILE_PROTEIN, 4, ENTRY_DIST_OCCUPANCY, 0.01f, -1) == RC_OK);
    CHECK(open_header(0xC6F1, PROFILE_PROTEIN, 4, ENTRY_DIST_OCCUPANCY, 0.01f, -1) == RC_EINVAL);
    CHECK(open_header(MAGIC_NUMBER, PROFILE_STANDARD, 4, ENTRY_DIST_OCCUPANCY, 0.01f, -1) == RC_EINVAL);
    CHECK(open_header(MAGIC_NUMBER, PROFILE_PROTEIN, 8, ENTRY_DIST_OCCUPANCY, 0.01f, -1) == RC_EINVAL);
    CHECK(open_header(MAGIC_NUMBER, PROFILE_PROTEIN, 4, ENTRY_DIST_NULL, 0.01f, -1) == RC_EINVAL);
    CHECK(open_header(MAGIC_NUMBER, PROFILE_PROTEIN, 4, 3, 0.01f, -1) == RC_EINVAL);
    CHECK(open_header(MAGIC_NUMBER, PROFILE_PROTEIN, 4, ENTRY_DIST_UNIFORM, 1.5f, -1) == RC_EINVAL);
    CHECK(open_header(MAGIC_NUMBER, PROFILE_PROTEIN, 4, ENTRY_DIST_UNIFORM, -0.1f, -1) == RC_EINVAL);
    CHECK(open_header(MAGIC_NUMBER, PROFILE_PROTEIN, 4, ENTRY_DIST_UNIFORM, 0.01f, 0) == RC_EIO);
}

/* N2 framing reader: a profile map(16) whose "null" / "alt" values are some other serialisation
 * (here: nested maps with arrays, strings, bins and ext objects, as imm's would be) is parsed and
 * validated up to the transitions, which are reported as unreadable -- RC_EPARSE, not a crash and
 * not a silently wrong profile; a dp value that is one key short of this library's is foreign too. */
static void write_foreign_dp(struct lip_file *f, unsigned depth)
{
    lip_write_map_size(f, 3);
    lip_write_cstr(f, "emis_score"), lip_write_1darray_size_type(f, 6, LIP_1DARRAY_F32);
    float v[6] = {0.5f, -1, 2, 3, 4, 5};
    lip_write_1darray_f32_data(f, 6, v);
    lip_write_cstr(f, "state_ids"), lip_write_array_size(f, 3), lip_write_int(f, 1), lip_write_int(f, 70000),
        lip_write_cstr(f, "x");
    lip_write_cstr(f, "nested");
    if (depth) write_foreign_dp(f, depth - 1);
    else lip_write_int(f, 0);
}

static enum rc unpack_with_foreign_dp(int foreign, unsigned core_size_field, unsigned nmatch)
{
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, imm_super(&imm_dna_iupac));
    struct protein_profile src, dst;
    protein_profile_init(&src, "src", &imm_amino_iupac, &code, PROTEIN_CFG_DEFAULT);
    protein_profile_sample(&src, 9, 3);
    char path[64];
    FILE *fp = tmp(path);
    struct lip_file f;
    lip_file_init(&f, fp);
    lip_write_map_size(&f, 16);
    lip_write_cstr(&f, "accession"), lip_write_cstr(&f, "PF99999.1");
    lip_write_cstr(&f, "null");
    if (foreign) write_foreign_dp(&f, 2);
    else imm_dp_pack(&src.null.dp, &f);
    lip_write_cstr(&f, "alt");
    if (foreign) write_foreign_dp(&f, 5);
    else imm_dp_pack(&src.alt.dp, &f);
    lip_write_cstr(&f, "core_size"), lip_write_int(&f, core_size_field);
    lip_write_cstr(&f, "consensus"), lip_write_cstr(&f, "ABC");
    char const *keys[8] = {"R", "S", "N", "B", "E", "J", "C", "T"};
    for (unsigned i = 0; i < 8; ++i)
        lip_write_cstr(&f, keys[i]), lip_write_int(&f, i ? i - 1 : 0);
    lip_write_cstr(&f, "null_ndist"), nuclt_dist_pack(&src.null.ndist, &f);
    lip_write_cstr(&f, "alt_insert_ndist"), nuclt_dist_pack(&src.alt.insert_ndist, &f);
    lip_write_cstr(&f, "alt_match_ndist"), lip_write_array_size(&f, nmatch);
    for (unsigned i = 0; i < nmatch; ++i)
        nuclt_dist_pack(&src.alt.match_ndists[i % 3], &f);
    rewind(fp);
    f.error = false;
    protein_profile_init(&dst, "", &imm_amino_iupac, &code, PROTEIN_CFG_DEFAULT);
    enum rc rc = profile_unpack(&dst.super, &f);
    if (!rc)
    {
        CHECK(!strcmp(dst.super.accession, "PF99999.1") && dst.core_size == 3 && !strcmp(dst.consensus, "ABC"));
        CHECK(!memcmp(dcp_profile_trans8(dst.impl), dcp_profile_trans8(src.impl), sizeof(float) * 24));
    }
    else
        CHECK(dst.impl == NULL); /* no half-built profile */
    /* the whole profile object was consumed either way: the stream is at its end */
    if (rc == RC_OK || foreign) CHECK(fgetc(fp) == EOF);
    profile_del(&dst.super);
    profile_del(&src.super);
    fclose(fp);
    remove(path);
    return rc;
}

static void foreign_dp(void)
{
    CHECK(unpack_with_foreign_dp(0, 3, 3) == RC_OK);
    CHECK(unpack_with_foreign_dp(1, 3, 3) == RC_EPARSE);
    CHECK(unpack_with_foreign_dp(0, 3, 2) == RC_EPARSE);    /* alt_match_ndist shorter than core_size */
    CHECK(unpack_with_foreign_dp(0, 4, 4) == RC_EPARSE);    /* dp value does not match core_size */
    CHECK(unpack_with_foreign_dp(0, 0, 0) == RC_EIO);       /* empty core */
    CHECK(unpack_with_foreign_dp(0, 5000, 3) == RC_EIO);    /* "profile is too long" */
}

static void lip_primitives(void)
{
    char path[64];
    FILE *fp = tmp(path);
    struct lip_file f;
    lip_file_init(&f, fp);
    uint64_t const ints[] = {0, 1, 127, 128, 255, 256, 65535, 65536, 0xffffffffu, 0x100000000ull, UINT64_MAX};
    for (unsigned i = 0; i < sizeof ints / sizeof ints[0]; ++i)
        CHECK(lip_write_uint(&f, ints[i]));
    CHECK(lip_write_f32(&f, -INFINITY) && lip_write_f32(&f, 0.1f));
    char longstr[300];
    memset(longstr, 'a', sizeof longstr - 1);
    longstr[sizeof longstr - 1] = '\0';
    CHECK(lip_write_cstr(&f, "") && lip_write_cstr(&f, "thirty-one chars long string...") && lip_write_cstr(&f, longstr));
    CHECK(lip_write_map_size(&f, 15) && lip_write_map_size(&f, 16) && lip_write_array_size(&f, 70000));
    rewind(fp);
    for (unsigned i = 0; i < sizeof ints / sizeof ints[0]; ++i)
    {
        uint64_t v = 1234;
        CHECK(lip_read_uint(&f, &v) && v == ints[i]);
    }
    float a = 0, b = 0;
    CHECK(lip_read_f32(&f, &a) && isinf(a) && a < 0 && lip_read_f32(&f, &b) && b == 0.1f);
    char buf[400];
    CHECK(lip_read_cstr(&f, sizeof buf, buf) && buf[0] == '\0');
    CHECK(lip_read_cstr(&f, sizeof buf, buf) && strlen(buf) == 31);
    CHECK(lip_read_cstr(&f, sizeof buf, buf) && !strcmp(buf, longstr));
    unsigned n = 0;
    CHECK(lip_read_map_size(&f, &n) && n == 15 && lip_read_map_size(&f, &n) && n == 16);
    CHECK(lip_read_array_size(&f, &n) && n == 70000);
    uint64_t past_end = 0;
    CHECK(!lip_read_uint(&f, &past_end) && f.error); /* end of file */
    /* a string that does not fit the buffer is an error, not a truncation */
    rewind(fp);
    f.error = false;
    CHECK(ftruncate(fileno(fp), 0) == 0);
    lip_write_cstr(&f, "0123456789");
    rewind(fp);
    CHECK(!lip_read_cstr(&f, 10, buf) && f.error);
    fclose(fp);
    remove(path);
}

/* products without any device call: prod_fwrite's row format (src/server/prod.c:13-41,153-181), the
 * caller's write_match_func once per step with ';' between steps, on-demand opening of a thread's file,
 * prod_fclose = header + rows in thread order */
static enum rc step_len_func(FILE *fp, void const *match)
{
    struct match const *m = match;
    return fprintf(fp, "%.*s/%u", (int)m->frag->size, m->frag->str, (unsigned)m->step->seqlen) < 0 ? RC_EIO : RC_OK;
}

static void products(void)
{
    struct prod p = {0};
    prod_setup_job(&p, "dna", "protein", 5);
    prod_setup_seq(&p, 7);
    snprintf(p.profile_name, sizeof p.profile_name, "PF1");
    p.alt_loglik = -54.35543441772461;
    p.null_loglik = -48.927268981933594;
    struct imm_step st[4] = {{PROTEIN_S_STATE, 0}, {PROTEIN_N_STATE, 3}, {PROTEIN_N_STATE, 1}, {PROTEIN_T_STATE, 0}};
    struct imm_path path = {st, 4, 4};
    struct imm_seq seq = imm_seq(imm_str("ACGT"), imm_super(imm_super(&imm_dna_iupac)));
    struct match m;
    match_setup(&m, NULL);
    CHECK(prod_fwrite(&p, &seq, &path, 3, step_len_func, &m) == RC_OK); /* threads 0..2 never opened */
    prod_setup_seq(&p, 8);
    CHECK(prod_fwrite(&p, &seq, &path, 1, step_len_func, &m) == RC_OK);
    struct imm_path too_long = {st, 4, 4};
    struct imm_seq shorter = imm_subseq(&seq, 0, 3);
    CHECK(prod_fwrite(&p, &shorter, &too_long, 1, step_len_func, &m) == RC_EINVAL); /* path longer than the sequence */
    CHECK(prod_fclose() == RC_OK);
    char buf[1024] = {0};
    CHECK(fread(buf, 1, sizeof buf - 1, prod_final_fp()) > 0);
    char const *want_rows = "5\t8\tPF1\tdna\t-54.355434417724609\t-48.927268981933594\tprotein\t" DECIPHON_VERSION "\t/0;ACG/3;T/1;/0\n";
    CHECK(!strncmp(buf, prod_header(), strlen(prod_header())));
    CHECK(!strncmp(buf + strlen(prod_header()), want_rows, strlen(want_rows))); /* thread 1 before thread 3 */
    CHECK(strstr(buf, "5\t7\tPF1\tdna\t") > strstr(buf, "5\t8\tPF1\tdna\t"));
    CHECK(access(prod_final_path(), R_OK) == 0);
    char path_copy[64];
    snprintf(path_copy, sizeof path_copy, "%s", prod_final_path());
    prod_final_cleanup();
    CHECK(access(path_copy, R_OK) != 0 && prod_final_fp() == NULL);
}

int main(void)
{
    products();
    lip_primitives();
    header_checks();
    roundtrip();
    foreign_dp();
    if (failed) fprintf(stderr, "%d check(s) failed\n", failed);
    else puts("test_db_host: all checks passed");
    return failed;
}
