/*
 * dcp_db_host.c -- C11 host layer, part 2: the pressed-database file (".dcp") and the partitioned
 * profile reader (include/deciphon_host.h section "db").
 *
 * File layout = the reference's (src/db/writer.c:95-117, src/db/protein_writer.c:56-96,
 * src/db/reader.c:25-79, src/db/protein_reader.c:40-82, file-format.md): a MessagePack
 *   map(2) { "header":   map(8) { magic_number, profile_typeid, float_size, entry_dist, epsilon,
 *                                 abc, amino, profile_sizes (1darray u32) },
 *            "profiles": array(N) of the map(16) of protein_profile_pack }
 * with the keys in exactly this order (the reference reads them positionally).  The values imm
 * serialises itself (abc, amino, the two dp of a profile) are in this library's own encoding: see
 * deciphon_host.c.  A file pressed by the reference therefore parses up to each profile's dp
 * values and stops there with RC_EPARSE.
 */
#include "deciphon_host.h"
#include "host_internal.h"

#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define fail dcp_host_fail

/* ---- db_reader (src/db/reader.c) --------------------------------------------------------------------- */
enum rc db_reader_open(struct db_reader *db, FILE *fp)
{
    db->nprofiles = 0;
    db->profile_sizes = NULL;
    db->profile_typeid = PROFILE_NULL;
    lip_file_init(&db->file, fp);
    return RC_OK;
}

void db_reader_close(struct db_reader *db)
{
    free(db->profile_sizes);
    db->profile_sizes = NULL;
}

enum rc db_reader_unpack_magic_number(struct db_reader *db)
{
    if (!expect_map_key(&db->file, "magic_number")) return fail(RC_EIO, "read key");
    unsigned number = 0;
    if (!lip_read_unsigned(&db->file, &number)) return fail(RC_EIO, "read magic number");
    return number != MAGIC_NUMBER ? fail(RC_EINVAL, "invalid magic number") : RC_OK;
}

enum rc db_reader_unpack_profile_typeid(struct db_reader *db, enum profile_typeid typeid)
{
    if (!expect_map_key(&db->file, "profile_typeid")) return fail(RC_EIO, "read key");
    unsigned v = 0;
    if (!lip_read_unsigned(&db->file, &v)) return fail(RC_EIO, "read typeid");
    db->profile_typeid = (enum profile_typeid)v;
    if (db->profile_typeid != typeid) return fail(RC_EINVAL, "invalid typeid");
    return RC_OK;
}

enum rc db_reader_unpack_float_size(struct db_reader *db)
{
    if (!expect_map_key(&db->file, "float_size")) return fail(RC_EIO, "read key");
    unsigned size = 0;
    if (!lip_read_unsigned(&db->file, &size)) return fail(RC_EIO, "read float size");
    return size != IMM_FLOAT_BYTES ? fail(RC_EINVAL, "invalid float size") : RC_OK;
}

enum rc db_reader_unpack_profile_sizes(struct db_reader *db)
{
    if (!expect_map_key(&db->file, "profile_sizes")) return fail(RC_EIO, "read key");
    enum lip_1darray_type type = 0;
    if (!lip_read_1darray_size_type(&db->file, &db->nprofiles, &type)) return fail(RC_EIO, "read array");
    if (type != LIP_1DARRAY_UINT32) return fail(RC_EINVAL, "invalid type");
    if (db->nprofiles > MAX_NPROFILES) return fail(RC_EINVAL, "too many profiles");
    db->profile_sizes = malloc(sizeof *db->profile_sizes * (db->nprofiles ? db->nprofiles : 1));
    if (!db->profile_sizes) return fail(RC_ENOMEM, "allocate memory");
    if (!lip_read_1darray_u32_data(&db->file, db->nprofiles, db->profile_sizes))
    {
        free(db->profile_sizes);
        db->profile_sizes = NULL;
        return fail(RC_EIO, "read array");
    }
    return RC_OK;
}

/* ---- protein_db_reader (src/db/protein_reader.c:40-82) ------------------------------------------------- */
enum rc protein_db_reader_open(struct protein_db_reader *db, FILE *fp)
{
    enum rc rc = db_reader_open(&db->super, fp);
    if (rc) return rc;
    struct lip_file *file = &db->super.file;
    if (!expect_map_size(file, 2)) return fail(RC_EIO, "read map");
    if (!expect_map_key(file, "header")) return fail(RC_EIO, "read key");
    if (!expect_map_size(file, 8)) return fail(RC_EIO, "read map");

    if ((rc = db_reader_unpack_magic_number(&db->super))) goto cleanup;
    if ((rc = db_reader_unpack_profile_typeid(&db->super, PROFILE_PROTEIN))) goto cleanup;
    if ((rc = db_reader_unpack_float_size(&db->super))) goto cleanup;

    unsigned edist = 0;
    if (!expect_map_key(file, "entry_dist") || !lip_read_unsigned(file, &edist))
    {
        rc = fail(RC_EIO, "read entry dist");
        goto cleanup;
    }
    if (edist <= ENTRY_DIST_NULL || edist > ENTRY_DIST_OCCUPANCY)
    {
        rc = fail(RC_EINVAL, "invalid entry dist");
        goto cleanup;
    }
    db->cfg.entry_dist = (enum entry_dist)edist;
    if (!expect_map_key(file, "epsilon") || !lip_read_f32(file, &db->cfg.epsilon))
    {
        rc = fail(RC_EIO, "read epsilon");
        goto cleanup;
    }
    if (!(db->cfg.epsilon >= 0 && db->cfg.epsilon <= 1))
    {
        rc = fail(RC_EINVAL, "invalid epsilon");
        goto cleanup;
    }
    if (!expect_map_key(file, "abc") || imm_abc_unpack(&db->nuclt.super, file))
    {
        rc = fail(RC_EIO, "read nuclt");
        goto cleanup;
    }
    if (!expect_map_key(file, "amino") || imm_abc_unpack(&db->amino.super, file))
    {
        rc = fail(RC_EIO, "read amino");
        goto cleanup;
    }
    /* the engine is built for 4-letter nucleotide and 20-letter amino alphabets */
    if (db->nuclt.super.size != IMM_NUCLT_SIZE || db->amino.super.size != IMM_AMINO_SIZE)
    {
        rc = fail(RC_EINVAL, "unsupported alphabet sizes %u / %u", db->nuclt.super.size, db->amino.super.size);
        goto cleanup;
    }
    imm_nuclt_code_init(&db->code, &db->nuclt);
    if ((rc = db_reader_unpack_profile_sizes(&db->super))) goto cleanup;
    return rc;

cleanup:
    db_reader_close(&db->super);
    return rc;
}

/* ---- db_writer (src/db/writer.c): header items and profiles go to temporary files first, because the
 * header's map size and profile_sizes are only known at close ------------------------------------------- */
static void destroy_tempfiles(struct db_writer *db)
{
    if (db->tmp.header.fp) fclose(db->tmp.header.fp);
    if (db->tmp.profile_sizes.fp) fclose(db->tmp.profile_sizes.fp);
    if (db->tmp.profiles.fp) fclose(db->tmp.profiles.fp);
    db->tmp.header.fp = db->tmp.profile_sizes.fp = db->tmp.profiles.fp = NULL;
}

static enum rc copy_stream(FILE *dst, FILE *src)
{
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, src)) > 0)
        if (fwrite(buf, 1, n, dst) != n) return fail(RC_EIO, "failed to write");
    return ferror(src) ? fail(RC_EIO, "failed to read") : RC_OK;
}

enum rc db_writer_open(struct db_writer *db, FILE *fp)
{
    db->nprofiles = 0;
    db->header_size = 0;
    lip_file_init(&db->file, fp);
    lip_file_init(&db->tmp.header, tmpfile());
    lip_file_init(&db->tmp.profile_sizes, tmpfile());
    lip_file_init(&db->tmp.profiles, tmpfile());
    if (!db->tmp.header.fp || !db->tmp.profile_sizes.fp || !db->tmp.profiles.fp)
    {
        destroy_tempfiles(db);
        return fail(RC_EIO, "create tmpfile");
    }
    return RC_OK;
}

enum rc db_writer_pack_magic_number(struct db_writer *db)
{
    if (!lip_write_cstr(&db->tmp.header, "magic_number") || !lip_write_int(&db->tmp.header, MAGIC_NUMBER))
        return fail(RC_EIO, "write magic number");
    db->header_size++;
    return RC_OK;
}

enum rc db_writer_pack_profile_typeid(struct db_writer *db, int profile_typeid)
{
    if (!lip_write_cstr(&db->tmp.header, "profile_typeid") || !lip_write_int(&db->tmp.header, profile_typeid))
        return fail(RC_EIO, "write profile_typeid");
    db->header_size++;
    return RC_OK;
}

enum rc db_writer_pack_float_size(struct db_writer *db)
{
    if (!lip_write_cstr(&db->tmp.header, "float_size") || !lip_write_int(&db->tmp.header, IMM_FLOAT_BYTES))
        return fail(RC_EIO, "write float size");
    db->header_size++;
    return RC_OK;
}

enum rc db_writer_pack_header_item(struct db_writer *db, pack_header_item_func_t pack_header_item, void const *arg)
{
    db->header_size++;
    return pack_header_item(&db->tmp.header, arg);
}

enum rc db_writer_pack_profile(struct db_writer *db, pack_profile_func_t pack_profile, void const *arg)
{
    long const start = ftell(db->tmp.profiles.fp);
    if (start < 0) return fail(RC_EIO, "ftell");
    enum rc rc = pack_profile(&db->tmp.profiles, arg);
    if (rc) return rc;
    long const end = ftell(db->tmp.profiles.fp);
    if (end < 0) return fail(RC_EIO, "ftell");
    if ((uint64_t)(end - start) > UINT32_MAX) return fail(RC_EFAIL, "profile is too large");
    if (!lip_write_int(&db->tmp.profile_sizes, (unsigned)(end - start))) return fail(RC_EIO, "write profile size");
    db->nprofiles++;
    return RC_OK;
}

enum rc db_writer_close(struct db_writer *db, bool successfully)
{
    enum rc rc = RC_OK;
    if (!successfully) goto cleanup;
    struct lip_file *file = &db->file;
    if (!lip_write_map_size(file, 2))
    {
        rc = fail(RC_EIO, "write root map size");
        goto cleanup;
    }
    /* "header": the items collected so far + profile_sizes */
    if (!lip_write_cstr(file, "header") || !lip_write_map_size(file, db->header_size + 1))
    {
        rc = fail(RC_EIO, "write header");
        goto cleanup;
    }
    rewind(db->tmp.header.fp);
    if ((rc = copy_stream(file->fp, db->tmp.header.fp))) goto cleanup;
    if (!lip_write_cstr(file, "profile_sizes") || !lip_write_1darray_size_type(file, db->nprofiles, LIP_1DARRAY_UINT32))
    {
        rc = fail(RC_EIO, "write profile sizes");
        goto cleanup;
    }
    rewind(db->tmp.profile_sizes.fp);
    db->tmp.profile_sizes.error = false;
    for (unsigned i = 0; i < db->nprofiles; ++i)
    {
        unsigned size = 0;
        if (!lip_read_unsigned(&db->tmp.profile_sizes, &size) || !lip_write_1darray_u32_item(file, size))
        {
            rc = fail(RC_EIO, "write profile sizes");
            goto cleanup;
        }
    }
    /* "profiles" */
    if (!lip_write_cstr(file, "profiles") || !lip_write_array_size(file, db->nprofiles))
    {
        rc = fail(RC_EIO, "write profiles");
        goto cleanup;
    }
    rewind(db->tmp.profiles.fp);
    if ((rc = copy_stream(file->fp, db->tmp.profiles.fp))) goto cleanup;
    if (fflush(file->fp)) rc = fail(RC_EIO, "failed to flush");

cleanup:
    destroy_tempfiles(db);
    return rc;
}

/* ---- protein_db_writer (src/db/protein_writer.c) -------------------------------------------------------- */
static enum rc pack_entry_dist_cb(struct lip_file *file, void const *arg)
{
    return lip_write_cstr(file, "entry_dist") && lip_write_int(file, *(enum entry_dist const *)arg)
               ? RC_OK
               : fail(RC_EIO, "write entry dist");
}
static enum rc pack_epsilon_cb(struct lip_file *file, void const *arg)
{
    return lip_write_cstr(file, "epsilon") && lip_write_float(file, *(imm_float const *)arg)
               ? RC_OK
               : fail(RC_EIO, "write epsilon");
}
static enum rc pack_nuclt_cb(struct lip_file *file, void const *arg)
{
    return lip_write_cstr(file, "abc") && !imm_abc_pack(&((struct imm_nuclt const *)arg)->super, file)
               ? RC_OK
               : fail(RC_EIO, "write nuclt abc");
}
static enum rc pack_amino_cb(struct lip_file *file, void const *arg)
{
    return lip_write_cstr(file, "amino") && !imm_abc_pack(&((struct imm_amino const *)arg)->super, file)
               ? RC_OK
               : fail(RC_EIO, "write amino abc");
}

enum rc protein_db_writer_open(struct protein_db_writer *db, FILE *fp, struct imm_amino const *amino,
                               struct imm_nuclt const *nuclt, struct protein_cfg cfg)
{
    enum rc rc = db_writer_open(&db->super, fp);
    if (rc) return rc;
    db->amino = *amino;
    db->nuclt = *nuclt;
    imm_nuclt_code_init(&db->code, &db->nuclt);
    db->cfg = cfg;
    if ((rc = db_writer_pack_magic_number(&db->super))) goto cleanup;
    if ((rc = db_writer_pack_profile_typeid(&db->super, PROFILE_PROTEIN))) goto cleanup;
    if ((rc = db_writer_pack_float_size(&db->super))) goto cleanup;
    if ((rc = db_writer_pack_header_item(&db->super, pack_entry_dist_cb, &db->cfg.entry_dist))) goto cleanup;
    if ((rc = db_writer_pack_header_item(&db->super, pack_epsilon_cb, &db->cfg.epsilon))) goto cleanup;
    if ((rc = db_writer_pack_header_item(&db->super, pack_nuclt_cb, &db->nuclt))) goto cleanup;
    if ((rc = db_writer_pack_header_item(&db->super, pack_amino_cb, &db->amino))) goto cleanup;
    return rc;

cleanup:
    db_writer_close(&db->super, false);
    return rc;
}

static enum rc pack_profile_cb(struct lip_file *file, void const *prof) { return protein_profile_pack(prof, file); }

enum rc protein_db_writer_pack_profile(struct protein_db_writer *db, struct protein_profile const *profile)
{
    /* one (entry_dist, epsilon) per database: the header carries them, the profiles do not */
    if (profile->cfg.entry_dist != db->cfg.entry_dist || profile->cfg.epsilon != db->cfg.epsilon)
        return fail(RC_EINVAL, "profile cfg differs from the database's");
    return db_writer_pack_profile(&db->super, pack_profile_cb, profile);
}

/* ---- profile_reader (src/db/profile_reader.c) ---------------------------------------------------------------
 * Every partition reads through its own FILE* on the same file (the reference re-opens the file from
 * the descriptor's path, xfile_open_from_fptr; here /proc/self/fd/N, falling back to a dup). */
static FILE *reopen(FILE *fp)
{
    char path[64];
    snprintf(path, sizeof path, "/proc/self/fd/%d", fileno(fp));
    FILE *f = fopen(path, "rb");
    if (f) return f;
    int fd = dup(fileno(fp));
    if (fd < 0) return NULL;
    f = fdopen(fd, "rb");
    if (!f) close(fd);
    return f;
}

static void close_files(struct profile_reader *reader)
{
    for (unsigned i = 0; i < reader->npartitions; ++i)
        if (reader->file[i].fp)
        {
            fclose(reader->file[i].fp);
            reader->file[i].fp = NULL;
        }
}

static enum rc reader_setup(struct profile_reader *reader, struct db_reader *db, unsigned npartitions, bool by_bytes)
{
    if (npartitions == 0) return fail(RC_EINVAL, "can't have zero partitions");
    if (npartitions > NUM_THREADS) return fail(RC_EINVAL, "too many partitions");
    memset(reader, 0, sizeof *reader);
    unsigned const nparts = xmath_min(npartitions, db->nprofiles);

    if (!expect_map_key(&db->file, "profiles")) return fail(RC_EIO, "read key");
    unsigned n = 0;
    if (!lip_read_array_size(&db->file, &n)) return fail(RC_EIO, "read array size");
    if (n != db->nprofiles) return fail(RC_EINVAL, "invalid nprofiles");
    long const profiles_offset = ftell(db->file.fp);
    if (profiles_offset < 0) return fail(RC_EIO, "ftell");

    reader->profile_typeid = db->profile_typeid;
    if (reader->profile_typeid != PROFILE_PROTEIN) return fail(RC_EINVAL, "only protein profiles can be scanned");
    struct protein_db_reader *pdb = (struct protein_db_reader *)db;
    for (unsigned i = 0; i < nparts; ++i)
    {
        FILE *f = reopen(db->file.fp);
        if (!f)
        {
            reader->npartitions = i;
            close_files(reader);
            reader->npartitions = 0;
            return fail(RC_EIO, "failed to open file");
        }
        lip_file_init(reader->file + i, f);
        protein_profile_init(&reader->profiles[i].pro, "", &pdb->amino, &pdb->code, pdb->cfg);
    }
    reader->npartitions = nparts;

    /* partition_init + partition_it (profile_reader.c:45-72): sizes, then byte offsets from
     * profile_sizes[].  ceil-sized partitions can leave trailing empty ones whose end offset the
     * reference never writes (stays 0): reproduced as is. */
    reader->partition_offset[0] = profiles_offset;
    if (!by_bytes)
    {
        unsigned i = 0, size = 0;
        for (unsigned j = 0; j < db->nprofiles; ++j)
        {
            reader->partition_offset[i + 1] += db->profile_sizes[j];
            if (++size >= xmath_partition_size(db->nprofiles, nparts, i))
            {
                reader->partition_size[i] = size;
                reader->partition_offset[i + 1] += reader->partition_offset[i];
                ++i;
                size = 0;
            }
        }
    }
    else
    {
        /* contiguous, every partition non-empty, boundaries at the nearest multiple of total / nparts */
        uint64_t total = 0;
        for (unsigned j = 0; j < db->nprofiles; ++j)
            total += db->profile_sizes[j];
        uint64_t acc = 0;
        unsigned i = 0, size = 0;
        for (unsigned j = 0; j < db->nprofiles; ++j)
        {
            acc += db->profile_sizes[j];
            ++size;
            unsigned const left_profiles = db->nprofiles - j - 1, left_parts = nparts - i - 1;
            bool const must_close = left_profiles == left_parts; /* one profile per remaining partition */
            bool const want_close = i + 1 < nparts && acc * nparts >= total * (i + 1);
            if (j + 1 == db->nprofiles || must_close || (want_close && left_profiles >= left_parts))
            {
                reader->partition_size[i] = size;
                reader->partition_offset[i + 1] = profiles_offset + (int64_t)acc;
                ++i;
                size = 0;
                if (i == nparts) break;
            }
        }
    }
    for (unsigned i = 0; i < nparts; ++i)
        reader->partition_first[i + 1] = reader->partition_first[i] + reader->partition_size[i];
    enum rc rc = profile_reader_rewind_all(reader);
    if (rc) close_files(reader);
    return rc;
}

enum rc profile_reader_setup(struct profile_reader *reader, struct db_reader *db, unsigned npartitions)
{
    return reader_setup(reader, db, npartitions, false);
}

enum rc profile_reader_setup_balanced(struct profile_reader *reader, struct db_reader *db, unsigned npartitions)
{
    return reader_setup(reader, db, npartitions, true);
}

unsigned profile_reader_npartitions(struct profile_reader const *reader) { return reader->npartitions; }

unsigned profile_reader_partition_size(struct profile_reader const *reader, unsigned partition)
{
    return reader->partition_size[partition];
}

unsigned profile_reader_nprofiles(struct profile_reader const *reader)
{
    unsigned n = 0;
    for (unsigned i = 0; i < reader->npartitions; ++i)
        n += reader->partition_size[i];
    return n;
}

enum rc profile_reader_rewind(struct profile_reader *reader, unsigned partition)
{
    struct lip_file *f = reader->file + partition;
    f->error = false;
    if (fseek(f->fp, (long)reader->partition_offset[partition], SEEK_SET)) return fail(RC_EIO, "failed to fseek");
    return RC_OK;
}

enum rc profile_reader_rewind_all(struct profile_reader *reader)
{
    for (unsigned i = 0; i < reader->npartitions; ++i)
    {
        enum rc rc = profile_reader_rewind(reader, i);
        if (rc) return rc;
    }
    return RC_OK;
}

enum rc profile_reader_next(struct profile_reader *reader, unsigned partition, struct profile **profile)
{
    *profile = (struct profile *)&reader->profiles[partition]; /* borrowed: overwritten by the next call */
    long const offset = ftell(reader->file[partition].fp);
    if (offset < 0) return fail(RC_EIO, "failed to ftello");
    /* an empty partition is at its end from the start (the reference would never match its unwritten
     * end offset 0 and read on into the next partition's profiles: profile_reader.c:140-146) */
    if (offset == reader->partition_offset[partition + 1] || reader->partition_size[partition] == 0) return RC_END;
    return profile_unpack(*profile, &reader->file[partition]);
}

bool profile_reader_end(struct profile_reader *reader, unsigned partition)
{
    (void)reader;
    (void)partition;
    return true; /* as the reference (profile_reader.c:170-175) */
}

void profile_reader_del(struct profile_reader *reader)
{
    for (unsigned i = 0; i < reader->npartitions; ++i)
        profile_del((struct profile *)&reader->profiles[i]);
    close_files(reader);
    reader->npartitions = 0;
}
