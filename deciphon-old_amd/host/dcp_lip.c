/*
 * dcp_lip.c -- the MessagePack stream calls deciphon's db / model code makes
 * (include/deciphon/core/lite_pack.h -> EBI-Metagenomics/lite-pack 0.3.0, absent here).
 *
 * lite-pack is a thin MessagePack reader/writer over FILE*; this is an own implementation of the
 * published MessagePack format for the calls on the scan path: maps, arrays, strings, unsigned
 * integers in the smallest form, float32, and "1darray" = an ext object whose type byte is the
 * element type and whose payload is the big-endian elements.  Readers accept every standard
 * encoding of a value of the right family (e.g. a uint written as uint8/16/32/64 or positive
 * fixint; a float written as float32 or float64) so that files from another MessagePack writer of
 * the same schema load; anything else sets file->error.
 * Call sites followed: src/db/reader.c:25-79, src/db/writer.c:45-165, src/db/protein_reader.c:8-82,
 * src/model/protein_profile.c:38-117,338-400, src/model/nuclt_dist.c:5-23, src/core/expect.c.
 */
#include "deciphon_host.h"

#include <string.h>

/* ---- raw bytes ------------------------------------------------------------------------------ */
static bool put(struct lip_file *f, void const *p, size_t n)
{
    if (f->error) return false;
    if (n && fwrite(p, 1, n, f->fp) != n) f->error = true;
    return !f->error;
}

static bool get(struct lip_file *f, void *p, size_t n)
{
    if (f->error) return false;
    if (n && fread(p, 1, n, f->fp) != n) f->error = true;
    return !f->error;
}

static bool put_be(struct lip_file *f, uint64_t v, unsigned nbytes)
{
    unsigned char b[8];
    for (unsigned i = 0; i < nbytes; ++i)
        b[i] = (unsigned char)(v >> (8 * (nbytes - 1 - i)));
    return put(f, b, nbytes);
}

static bool get_be(struct lip_file *f, uint64_t *v, unsigned nbytes)
{
    unsigned char b[8];
    if (!get(f, b, nbytes)) return false;
    uint64_t x = 0;
    for (unsigned i = 0; i < nbytes; ++i)
        x = (x << 8) | b[i];
    *v = x;
    return true;
}

static bool fail(struct lip_file *f)
{
    f->error = true;
    return false;
}

/* ---- writers ---------------------------------------------------------------------------------- */
static bool put_tag(struct lip_file *f, unsigned char tag) { return put(f, &tag, 1); }

bool lip_write_map_size(struct lip_file *f, unsigned size)
{
    if (size < 16) return put_tag(f, (unsigned char)(0x80 | size));
    if (size <= 0xffff) return put_tag(f, 0xde) && put_be(f, size, 2);
    return put_tag(f, 0xdf) && put_be(f, size, 4);
}

bool lip_write_array_size(struct lip_file *f, unsigned size)
{
    if (size < 16) return put_tag(f, (unsigned char)(0x90 | size));
    if (size <= 0xffff) return put_tag(f, 0xdc) && put_be(f, size, 2);
    return put_tag(f, 0xdd) && put_be(f, size, 4);
}

bool lip_write_cstr(struct lip_file *f, char const *str)
{
    size_t const n = strlen(str);
    bool ok;
    if (n < 32) ok = put_tag(f, (unsigned char)(0xa0 | n));
    else if (n <= 0xff) ok = put_tag(f, 0xd9) && put_be(f, n, 1);
    else if (n <= 0xffff) ok = put_tag(f, 0xda) && put_be(f, n, 2);
    else if (n <= 0xffffffffu) ok = put_tag(f, 0xdb) && put_be(f, n, 4);
    else return fail(f);
    return ok && put(f, str, n);
}

bool lip_write_uint(struct lip_file *f, uint64_t v)
{
    if (v < 128) return put_tag(f, (unsigned char)v);
    if (v <= 0xff) return put_tag(f, 0xcc) && put_be(f, v, 1);
    if (v <= 0xffff) return put_tag(f, 0xcd) && put_be(f, v, 2);
    if (v <= 0xffffffffu) return put_tag(f, 0xce) && put_be(f, v, 4);
    return put_tag(f, 0xcf) && put_be(f, v, 8);
}

bool lip_write_f32(struct lip_file *f, float val)
{
    uint32_t bits;
    memcpy(&bits, &val, 4);
    return put_tag(f, 0xca) && put_be(f, bits, 4);
}

static unsigned elem_bytes(unsigned type)
{
    switch (type)
    {
    case LIP_1DARRAY_UINT8: return 1;
    case LIP_1DARRAY_UINT16: return 2;
    case LIP_1DARRAY_UINT32: return 4;
    case LIP_1DARRAY_F32: return 4;
    default: return 0;
    }
}

bool lip_write_1darray_size_type(struct lip_file *f, unsigned size, uint8_t type)
{
    unsigned const eb = elem_bytes(type);
    if (!eb) return fail(f);
    uint64_t const bytes = (uint64_t)size * eb;
    bool ok;
    if (bytes <= 0xff) ok = put_tag(f, 0xc7) && put_be(f, bytes, 1);
    else if (bytes <= 0xffff) ok = put_tag(f, 0xc8) && put_be(f, bytes, 2);
    else if (bytes <= 0xffffffffu) ok = put_tag(f, 0xc9) && put_be(f, bytes, 4);
    else return fail(f);
    return ok && put_tag(f, type);
}

bool lip_write_1darray_u32_item(struct lip_file *f, uint32_t item) { return put_be(f, item, 4); }

/* element data of the arrays is big-endian on file; moved in blocks (one stdio call per KiB, not per element:
 * a 20k-profile database holds half a billion floats) */
static bool put_be32_block(struct lip_file *f, unsigned size, void const *data)
{
    uint32_t chunk[256];
    unsigned char const *src = data;
    while (size)
    {
        unsigned const n = size < 256u ? size : 256u;
        memcpy(chunk, src, (size_t)n * 4);
        for (unsigned i = 0; i < n; ++i)
            chunk[i] = __builtin_bswap32(chunk[i]);
        if (!put(f, chunk, (size_t)n * 4)) return false;
        src += (size_t)n * 4, size -= n;
    }
    return true;
}

static bool get_be32_block(struct lip_file *f, unsigned size, void *data)
{
    if (!get(f, data, (size_t)size * 4)) return false;
    unsigned char *p = data;
    for (unsigned i = 0; i < size; ++i, p += 4)
    {
        uint32_t v;
        memcpy(&v, p, 4);
        v = __builtin_bswap32(v);
        memcpy(p, &v, 4);
    }
    return true;
}

bool lip_write_1darray_f32_data(struct lip_file *f, unsigned size, float const *data)
{
    return put_be32_block(f, size, data);
}

bool lip_write_1darray_u8_data(struct lip_file *f, unsigned size, uint8_t const *data) { return put(f, data, size); }

/* ---- readers ------------------------------------------------------------------------------------ */
static bool get_tag(struct lip_file *f, unsigned char *tag) { return get(f, tag, 1); }

static bool read_len(struct lip_file *f, unsigned char tag, unsigned char fix_mask, unsigned char fix_base,
                     unsigned char t8, unsigned char t16, unsigned char t32, unsigned *size)
{
    uint64_t v = 0;
    if (fix_mask && (tag & (unsigned char)~fix_mask) == fix_base) v = tag & fix_mask;
    else if (t8 && tag == t8)
    {
        if (!get_be(f, &v, 1)) return false;
    }
    else if (tag == t16)
    {
        if (!get_be(f, &v, 2)) return false;
    }
    else if (tag == t32)
    {
        if (!get_be(f, &v, 4)) return false;
    }
    else return fail(f);
    *size = (unsigned)v;
    return true;
}

bool lip_read_map_size(struct lip_file *f, unsigned *size)
{
    unsigned char tag;
    return get_tag(f, &tag) && read_len(f, tag, 0x0f, 0x80, 0, 0xde, 0xdf, size);
}

bool lip_read_array_size(struct lip_file *f, unsigned *size)
{
    unsigned char tag;
    return get_tag(f, &tag) && read_len(f, tag, 0x0f, 0x90, 0, 0xdc, 0xdd, size);
}

bool lip_read_str_size(struct lip_file *f, unsigned *size)
{
    unsigned char tag;
    return get_tag(f, &tag) && read_len(f, tag, 0x1f, 0xa0, 0xd9, 0xda, 0xdb, size);
}

bool lip_read_str_data(struct lip_file *f, unsigned size, char *str) { return get(f, str, size); }

bool lip_read_cstr(struct lip_file *f, unsigned size, char *str)
{
    unsigned n = 0;
    if (size == 0) return fail(f);
    str[0] = '\0';
    if (!lip_read_str_size(f, &n)) return false;
    if (n >= size) return fail(f);
    if (!get(f, str, n)) return false;
    str[n] = '\0';
    return true;
}

bool lip_read_uint(struct lip_file *f, uint64_t *val)
{
    unsigned char tag;
    if (!get_tag(f, &tag)) return false;
    if (tag < 128)
    {
        *val = tag;
        return true;
    }
    switch (tag)
    {
    case 0xcc: return get_be(f, val, 1);
    case 0xcd: return get_be(f, val, 2);
    case 0xce: return get_be(f, val, 4);
    case 0xcf: return get_be(f, val, 8);
    /* a non-negative value in a signed encoding */
    case 0xd0: case 0xd1: case 0xd2: case 0xd3:
    {
        unsigned const nb = 1u << (tag - 0xd0);
        uint64_t raw;
        if (!get_be(f, &raw, nb)) return false;
        if (raw >> (8 * nb - 1)) return fail(f); /* negative */
        *val = raw;
        return true;
    }
    default: return fail(f);
    }
}

bool lip_read_unsigned(struct lip_file *f, unsigned *val)
{
    uint64_t v;
    if (!lip_read_uint(f, &v)) return false;
    if (v > 0xffffffffu) return fail(f);
    *val = (unsigned)v;
    return true;
}

bool lip_read_int_as_int(struct lip_file *f, int *val)
{
    uint64_t v;
    if (!lip_read_uint(f, &v)) return false;
    if (v > 0x7fffffff) return fail(f);
    *val = (int)v;
    return true;
}

bool lip_read_f32(struct lip_file *f, float *val)
{
    unsigned char tag;
    uint64_t raw;
    if (!get_tag(f, &tag)) return false;
    if (tag == 0xca)
    {
        if (!get_be(f, &raw, 4)) return false;
        uint32_t bits = (uint32_t)raw;
        memcpy(val, &bits, 4);
        return true;
    }
    if (tag == 0xcb)
    {
        if (!get_be(f, &raw, 8)) return false;
        double d;
        memcpy(&d, &raw, 8);
        *val = (float)d;
        return true;
    }
    return fail(f);
}

bool lip_read_1darray_size_type(struct lip_file *f, unsigned *size, enum lip_1darray_type *type)
{
    unsigned char tag, ty;
    unsigned bytes = 0;
    if (!get_tag(f, &tag)) return false;
    if (!read_len(f, tag, 0, 0, 0xc7, 0xc8, 0xc9, &bytes)) return false;
    if (!get_tag(f, &ty)) return false;
    unsigned const eb = elem_bytes(ty);
    if (!eb || bytes % eb) return fail(f);
    *size = bytes / eb;
    *type = (enum lip_1darray_type)ty;
    return true;
}

bool lip_read_1darray_u32_data(struct lip_file *f, unsigned size, uint32_t *data)
{
    return get_be32_block(f, size, data);
}

bool lip_read_1darray_f32_data(struct lip_file *f, unsigned size, float *data)
{
    return get_be32_block(f, size, data);
}

bool lip_read_1darray_u8_data(struct lip_file *f, unsigned size, uint8_t *data) { return get(f, data, size); }

/* Skip one object of any type.  Iterative over a counter of pending objects, so a hostile depth
 * cannot exhaust the stack; payloads are skipped with fseek. */
bool lip_skip_object(struct lip_file *f)
{
    uint64_t pending = 1;
    while (pending)
    {
        unsigned char tag;
        uint64_t n = 0;
        if (!get_tag(f, &tag)) return false;
        --pending;
        uint64_t skip = 0;
        if (tag < 0x80 || tag >= 0xe0 || tag == 0xc0 || tag == 0xc2 || tag == 0xc3) skip = 0;
        else if ((tag & 0xf0) == 0x80) pending += 2ull * (tag & 0x0f);
        else if ((tag & 0xf0) == 0x90) pending += tag & 0x0f;
        else if ((tag & 0xe0) == 0xa0) skip = tag & 0x1f;
        else
            switch (tag)
            {
            case 0xc4: case 0xd9: if (!get_be(f, &n, 1)) return false; skip = n; break;
            case 0xc5: case 0xda: if (!get_be(f, &n, 2)) return false; skip = n; break;
            case 0xc6: case 0xdb: if (!get_be(f, &n, 4)) return false; skip = n; break;
            case 0xc7: if (!get_be(f, &n, 1)) return false; skip = n + 1; break;
            case 0xc8: if (!get_be(f, &n, 2)) return false; skip = n + 1; break;
            case 0xc9: if (!get_be(f, &n, 4)) return false; skip = n + 1; break;
            case 0xca: skip = 4; break;
            case 0xcb: skip = 8; break;
            case 0xcc: case 0xd0: skip = 1; break;
            case 0xcd: case 0xd1: skip = 2; break;
            case 0xce: case 0xd2: skip = 4; break;
            case 0xcf: case 0xd3: skip = 8; break;
            case 0xd4: skip = 2; break;
            case 0xd5: skip = 3; break;
            case 0xd6: skip = 5; break;
            case 0xd7: skip = 9; break;
            case 0xd8: skip = 17; break;
            case 0xdc: if (!get_be(f, &n, 2)) return false; pending += n; break;
            case 0xdd: if (!get_be(f, &n, 4)) return false; pending += n; break;
            case 0xde: if (!get_be(f, &n, 2)) return false; pending += 2 * n; break;
            case 0xdf: if (!get_be(f, &n, 4)) return false; pending += 2 * n; break;
            default: return fail(f); /* 0xc1: never used */
            }
        while (skip)
        {
            /* read, not seek: a truncated file must be noticed */
            char buf[4096];
            size_t const chunk = skip < sizeof buf ? (size_t)skip : sizeof buf;
            if (!get(f, buf, chunk)) return false;
            skip -= chunk;
        }
    }
    return true;
}

/* ---- include/deciphon/core/expect.h ------------------------------------------------------------------ */
bool expect_map_size(struct lip_file *file, unsigned size)
{
    unsigned sz = 0;
    if (!lip_read_map_size(file, &sz)) return false;
    return size == sz;
}

bool expect_map_key(struct lip_file *file, char const key[])
{
    unsigned size = 0;
    char buf[32] = {0};
    if (!lip_read_str_size(file, &size)) return false;
    if (size >= sizeof buf) return fail(file);
    if (!lip_read_str_data(file, size, buf)) return false;
    if (size != (unsigned)strlen(key) || strncmp(key, buf, size) != 0) return fail(file);
    return true;
}
