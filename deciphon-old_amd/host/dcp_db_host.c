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

#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define fail dcp_host_fail

/* ---- the header, as data -------------------------------------------------------------------------------
 * One table says what the header map holds -- key, wire type, where the value lives in a protein database
 * object, and the test it must pass.  ONE reader and ONE writer walk it; the per-field entry points of
 * the reference's API (db_reader_unpack_*, db_writer_pack_*: src/db/reader.c:25-79, src/db/writer.c:45-93)
 * are single-row walks.  Row order = file order (the reference reads the keys positionally). */
enum hdr_wire
{
    W_UINT,  /* MessagePack unsigned */
    W_F32,   /* float32 */
    W_ABC,   /* imm_abc value (map) */
    W_SIZES, /* 1darray of uint32: bytes of every profile */
};

enum hdr_row
{
    H_MAGIC,
    H_TYPEID,
    H_FLOAT_SIZE,
    H_ENTRY_DIST,
    H_EPSILON,
    H_NUCLT,
    H_AMINO,
    H_SIZES,
    H_ROWS
};

/* what the rows read into / write from */
struct hdr_values
{
    unsigned magic, typeid_, float_size, entry_dist;
    float epsilon;
    struct imm_abc *nuclt, *amino; /* only the rows W_ABC touch them */
    struct db_reader *reader;      /* W_SIZES fills nprofiles / profile_sizes */
    unsigned want_typeid;
};

static bool ok_magic(struct hdr_values const *v) { return v->magic == MAGIC_NUMBER; }
static bool ok_typeid(struct hdr_values const *v) { return v->typeid_ == v->want_typeid; }
static bool ok_float_size(struct hdr_values const *v) { return v->float_size == IMM_FLOAT_BYTES; }
static bool ok_entry_dist(struct hdr_values const *v)
{
    return v->entry_dist > ENTRY_DIST_NULL && v->entry_dist <= ENTRY_DIST_OCCUPANCY;
}
static bool ok_epsilon(struct hdr_values const *v) { return v->epsilon >= 0 && v->epsilon <= 1; }
static bool ok_nuclt(struct hdr_values const *v) { return v->nuclt->size == IMM_NUCLT_SIZE; }
static bool ok_amino(struct hdr_values const *v) { return v->amino->size == IMM_AMINO_SIZE; }
static bool ok_sizes(struct hdr_values const *v) { return v->reader->nprofiles <= MAX_NPROFILES; }

static struct
{
    char const *key;
    enum hdr_wire wire;
    size_t at; /* offset of the scalar in hdr_values (W_UINT, W_F32) */
    bool (*valid)(struct hdr_values const *);
} const hdr_schema[H_ROWS] = {
    [H_MAGIC] = {"magic_number", W_UINT, offsetof(struct hdr_values, magic), ok_magic},
    [H_TYPEID] = {"profile_typeid", W_UINT, offsetof(struct hdr_values, typeid_), ok_typeid},
    [H_FLOAT_SIZE] = {"float_size", W_UINT, offsetof(struct hdr_values, float_size), ok_float_size},
    [H_ENTRY_DIST] = {"entry_dist", W_UINT, offsetof(struct hdr_values, entry_dist), ok_entry_dist},
    [H_EPSILON] = {"epsilon", W_F32, offsetof(struct hdr_values, epsilon), ok_epsilon},
    [H_NUCLT] = {"abc", W_ABC, 0, ok_nuclt},
    [H_AMINO] = {"amino", W_ABC, 0, ok_amino},
    [H_SIZES] = {"profile_sizes", W_SIZES, 0, ok_sizes},
};

/* rows [first, last] from the file into *v: a key or value that cannot be read is RC_EIO, a value that
 * fails its row's test RC_EINVAL */
static enum rc hdr_read(struct lip_file *file, struct hdr_values *v, enum hdr_row first, enum hdr_row last)
{
    for (int r = first; r <= (int)last; ++r)
    {
        char const *key = hdr_schema[r].key;
        void *slot = (char *)v + hdr_schema[r].at;
        bool got = expect_map_key(file, key);
        if (got) switch (hdr_schema[r].wire)
            {
            case W_UINT: got = lip_read_unsigned(file, slot); break;
            case W_F32: got = lip_read_f32(file, slot); break;
            case W_ABC: got = imm_abc_unpack(r == H_NUCLT ? v->nuclt : v->amino, file) == IMM_OK; break;
            case W_SIZES:
            {
                struct db_reader *db = v->reader;
                enum lip_1darray_type ty = 0;
                got = lip_read_1darray_size_type(file, &db->nprofiles, &ty);
                if (got && ty != LIP_1DARRAY_UINT32) return fail(RC_EINVAL, "header field %s: not a uint32 array", key);
                if (got && !hdr_schema[r].valid(v)) return fail(RC_EINVAL, "header field %s: %u profiles is too many", key, db->nprofiles);
                if (got)
                {
                    db->profile_sizes = malloc(sizeof *db->profile_sizes * (db->nprofiles ? db->nprofiles : 1));
                    if (!db->profile_sizes) return fail(RC_ENOMEM, "header field %s: out of memory", key);
                    got = lip_read_1darray_u32_data(file, db->nprofiles, db->profile_sizes);
                    if (!got)
                    {
                        free(db->profile_sizes);
                        db->profile_sizes = NULL;
                    }
                }
                break;
            }
            }
        if (!got) return fail(RC_EIO, "header field %s: cannot be read", key);
        if (!hdr_schema[r].valid(v)) return fail(RC_EINVAL, "header field %s: value not accepted", key);
    }
    return RC_OK;
}

/* row r from *v to `out` (key, then value) */
static enum rc hdr_write(struct lip_file *out, struct hdr_values const *v, enum hdr_row r)
{
    void const *slot = (char const *)v + hdr_schema[r].at;
    bool ok = lip_write_cstr(out, hdr_schema[r].key);
    if (ok) switch (hdr_schema[r].wire)
        {
        case W_UINT: ok = lip_write_uint(out, *(unsigned const *)slot); break;
        case W_F32: ok = lip_write_f32(out, *(float const *)slot); break;
        case W_ABC: ok = imm_abc_pack(r == H_NUCLT ? v->nuclt : v->amino, out) == IMM_OK; break;
        case W_SIZES: ok = false; break; /* written by db_writer_close from the sizes it collected */
        }
    return ok ? RC_OK : fail(RC_EIO, "header field %s: cannot be written", hdr_schema[r].key);
}

/* ---- db_reader (include/deciphon/db/reader.h) -------------------------------------------------------- */
enum rc db_reader_open(struct db_reader *db, FILE *fp)
{
    *db = (struct db_reader){.nprofiles = 0, .profile_sizes = NULL, .profile_typeid = PROFILE_NULL};
    lip_file_init(&db->file, fp);
    return RC_OK;
}

void db_reader_close(struct db_reader *db)
{
    free(db->profile_sizes);
    db->profile_sizes = NULL;
}

static enum rc reader_row(struct db_reader *db, enum hdr_row r, unsigned want_typeid)
{
    struct hdr_values v = {.reader = db, .want_typeid = want_typeid};
    enum rc rc = hdr_read(&db->file, &v, r, r);
    if (r == H_TYPEID) db->profile_typeid = (enum profile_typeid)v.typeid_;
    return rc;
}
enum rc db_reader_unpack_magic_number(struct db_reader *db) { return reader_row(db, H_MAGIC, 0); }
enum rc db_reader_unpack_profile_typeid(struct db_reader *db, enum profile_typeid typeid) { return reader_row(db, H_TYPEID, typeid); }
enum rc db_reader_unpack_float_size(struct db_reader *db) { return reader_row(db, H_FLOAT_SIZE, 0); }
enum rc db_reader_unpack_profile_sizes(struct db_reader *db) { return reader_row(db, H_SIZES, 0); }

/* ---- protein_db_reader (include/deciphon/db/protein_reader.h): root map, then every row of the table --- */
enum rc protein_db_reader_open(struct protein_db_reader *db, FILE *fp)
{
    enum rc rc = db_reader_open(&db->super, fp);
    if (rc) return rc;
    struct lip_file *file = &db->super.file;
    if (!expect_map_size(file, 2) || !expect_map_key(file, "header") || !expect_map_size(file, H_ROWS))
        return fail(RC_EIO, "not a database: root map / header map");
    struct hdr_values v = {.nuclt = &db->nuclt.super, .amino = &db->amino.super, .reader = &db->super,
                           .want_typeid = PROFILE_PROTEIN};
    rc = hdr_read(file, &v, H_MAGIC, H_SIZES);
    db->super.profile_typeid = (enum profile_typeid)v.typeid_;
    if (rc)
    {
        db_reader_close(&db->super);
        return rc;
    }
    db->cfg = protein_cfg((enum entry_dist)v.entry_dist, v.epsilon);
    imm_nuclt_code_init(&db->code, &db->nuclt);
    return RC_OK;
}

/* ---- db_writer (include/deciphon/db/writer.h) ------------------------------------------------------------
 * The header's item count and profile_sizes are known only at close, so header items and sizes are
 * collected in memory streams (fmemopen: a few hundred bytes of header, 4 bytes per profile) and only the
 * profiles -- gigabytes for a Pfam-sized database -- in a temporary file.  The three lip_file members are
 * the reference's; a writer owns nothing outside them, so writers are independent of each other. */
enum
{
    HDR_STAGE_BYTES = 1 << 16,
    SIZES_STAGE_BYTES = 4 * (MAX_NPROFILES + 1),
};

static void writer_drop(struct db_writer *db)
{
    FILE **fps[3] = {&db->tmp.header.fp, &db->tmp.profile_sizes.fp, &db->tmp.profiles.fp};
    for (int i = 0; i < 3; ++i)
        if (*fps[i])
        {
            fclose(*fps[i]);
            *fps[i] = NULL;
        }
}

enum rc db_writer_open(struct db_writer *db, FILE *fp)
{
    db->nprofiles = db->header_size = 0;
    lip_file_init(&db->file, fp);
    lip_file_init(&db->tmp.header, fmemopen(NULL, HDR_STAGE_BYTES, "w+"));
    lip_file_init(&db->tmp.profile_sizes, fmemopen(NULL, SIZES_STAGE_BYTES, "w+"));
    lip_file_init(&db->tmp.profiles, tmpfile());
    if (db->tmp.header.fp && db->tmp.profile_sizes.fp && db->tmp.profiles.fp) return RC_OK;
    writer_drop(db);
    return fail(RC_EIO, "cannot create the writer's staging streams");
}

static enum rc writer_row(struct db_writer *db, enum hdr_row r, struct hdr_values const *v)
{
    enum rc rc = hdr_write(&db->tmp.header, v, r);
    if (!rc) db->header_size++;
    return rc;
}
enum rc db_writer_pack_magic_number(struct db_writer *db)
{
    return writer_row(db, H_MAGIC, &(struct hdr_values){.magic = MAGIC_NUMBER});
}
enum rc db_writer_pack_profile_typeid(struct db_writer *db, int profile_typeid)
{
    return writer_row(db, H_TYPEID, &(struct hdr_values){.typeid_ = (unsigned)profile_typeid});
}
enum rc db_writer_pack_float_size(struct db_writer *db)
{
    return writer_row(db, H_FLOAT_SIZE, &(struct hdr_values){.float_size = IMM_FLOAT_BYTES});
}
enum rc db_writer_pack_header_item(struct db_writer *db, pack_header_item_func_t pack_header_item, void const *arg)
{
    db->header_size++;
    return pack_header_item(&db->tmp.header, arg);
}

enum rc db_writer_pack_profile(struct db_writer *db, pack_profile_func_t pack_profile, void const *arg)
{
    FILE *fp = db->tmp.profiles.fp;
    long const before = ftell(fp);
    enum rc rc = before < 0 ? fail(RC_EIO, "profile staging file") : pack_profile(&db->tmp.profiles, arg);
    if (rc) return rc;
    long const after = ftell(fp);
    if (after < before) return fail(RC_EIO, "profile staging file");
    if ((uint64_t)(after - before) > UINT32_MAX) return fail(RC_EFAIL, "a profile of %ld bytes does not fit profile_sizes", after - before);
    uint32_t const bytes = (uint32_t)(after - before); /* raw: becomes one element of the 1darray at close */
    if (fwrite(&bytes, sizeof bytes, 1, db->tmp.profile_sizes.fp) != 1) return fail(RC_EIO, "profile size list");
    db->nprofiles++;
    return RC_OK;
}

/* the first `limit` bytes of `src` (SIZE_MAX: all of it) appended to `dst` */
static bool copy_bytes(FILE *dst, FILE *src, size_t limit)
{
    static _Thread_local char buf[1 << 16];
    rewind(src);
    while (limit)
    {
        size_t const want = limit < sizeof buf ? limit : sizeof buf, n = fread(buf, 1, want, src);
        if (n == 0) break;
        if (fwrite(buf, 1, n, dst) != n) return false;
        limit -= n;
    }
    return !ferror(src) && (limit == 0 || limit > ((size_t)-1) / 2);
}

enum rc db_writer_close(struct db_writer *db, bool successfully)
{
    enum rc rc = RC_OK;
    if (successfully)
    {
        struct lip_file *out = &db->file;
        long const hdr_bytes = ftell(db->tmp.header.fp);
        /* root: {"header": {items..., "profile_sizes": [...]}, "profiles": [...]} */
        bool ok = hdr_bytes >= 0 && lip_write_map_size(out, 2) && lip_write_cstr(out, "header") &&
                  lip_write_map_size(out, db->header_size + 1) && copy_bytes(out->fp, db->tmp.header.fp, (size_t)hdr_bytes);
        ok = ok && ftell(db->tmp.profile_sizes.fp) == (long)((size_t)db->nprofiles * sizeof(uint32_t)) &&
             lip_write_cstr(out, hdr_schema[H_SIZES].key) &&
             lip_write_1darray_size_type(out, db->nprofiles, LIP_1DARRAY_UINT32);
        if (ok) rewind(db->tmp.profile_sizes.fp);
        for (unsigned i = 0; ok && i < db->nprofiles; ++i)
        {
            uint32_t bytes;
            ok = fread(&bytes, sizeof bytes, 1, db->tmp.profile_sizes.fp) == 1 && lip_write_1darray_u32_item(out, bytes);
        }
        ok = ok && lip_write_cstr(out, "profiles") && lip_write_array_size(out, db->nprofiles) &&
             copy_bytes(out->fp, db->tmp.profiles.fp, (size_t)-1) && fflush(out->fp) == 0;
        if (!ok) rc = fail(RC_EIO, "cannot write the database file");
    }
    writer_drop(db);
    return rc;
}

/* ---- protein_db_writer (include/deciphon/db/protein_writer.h) ------------------------------------------- */
enum rc protein_db_writer_open(struct protein_db_writer *db, FILE *fp, struct imm_amino const *amino,
                               struct imm_nuclt const *nuclt, struct protein_cfg cfg)
{
    enum rc rc = db_writer_open(&db->super, fp);
    if (rc) return rc;
    db->amino = *amino;
    db->nuclt = *nuclt;
    imm_nuclt_code_init(&db->code, &db->nuclt);
    db->cfg = cfg;
    struct hdr_values const v = {.magic = MAGIC_NUMBER, .typeid_ = PROFILE_PROTEIN, .float_size = IMM_FLOAT_BYTES,
                                 .entry_dist = (unsigned)cfg.entry_dist, .epsilon = cfg.epsilon,
                                 .nuclt = &db->nuclt.super, .amino = &db->amino.super};
    for (int r = H_MAGIC; r < H_SIZES && !rc; ++r)
        rc = writer_row(&db->super, (enum hdr_row)r, &v);
    if (rc) db_writer_close(&db->super, false);
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
FILE *dcp_host_reopen(FILE *fp)
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
        FILE *f = dcp_host_reopen(db->file.fp);
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

    /* Count-balanced partitions (profile_reader.c:45-72, xmath.h:24-30): partition i takes the next
     * xmath_partition_size(n, nparts, i) profiles; its end offset is its start plus their bytes.  The
     * ceil-sized shares can exhaust the profiles early: the trailing partitions are empty and -- as in the
     * reference, which never reaches them in its loop -- their end offsets are never written (stay 0). */
    reader->partition_offset[0] = profiles_offset;
    if (!by_bytes)
    {
        unsigned next = 0;
        for (unsigned i = 0; i < nparts && next < db->nprofiles; ++i)
        {
            unsigned const share = xmath_partition_size(db->nprofiles, nparts, i);
            int64_t bytes = 0;
            for (unsigned j = next; j < next + share; ++j)
                bytes += db->profile_sizes[j];
            reader->partition_size[i] = share;
            reader->partition_offset[i + 1] = reader->partition_offset[i] + bytes;
            next += share;
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
    reader->profile_sizes = db->profile_sizes;
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
