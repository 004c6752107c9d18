/*
 * oracle/oracle_io.c -- CPU ORACLE, file readers (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Independent readers for the two file formats either side of the scan path, so that the parity
 * tests of SURVEY.md 8f N2 / N3 compare the HIP path with the oracle READING THE SAME BYTES instead of
 * comparing the product with itself:
 *
 *   orc_h3_*   HMMER3/f ASCII profiles -> oracle profiles, what protein_h3reader_next +
 *              protein_profile_absorb do in hmm_press (src/model/protein_h3reader.c:18-72,79-103;
 *              src/server/hmm.c:120-178).  The reference parses with the third-party `hmr` library
 *              (EBI-Metagenomics/hmmer-reader 0.1.3, absent): restated from the published HMMER3 text
 *              format; values are -ln p in decimal, '*' = probability 0, converted through double as
 *              hmr does and cast to imm_float (protein_h3reader.c:30-38,47-49).
 *   orc_dcp_*  the MessagePack database (".dcp"): map(2){header map(8), profiles array of map(16)} with
 *              the keys of src/db/writer.c:95-117, src/db/protein_reader.c:40-82 and
 *              src/model/protein_profile.c:338-400; nuclt_dist = array(2) of two float 1darrays
 *              (src/model/nuclt_dist.c:5-23).  The two dp values of a profile are the PRODUCT's own
 *              encoding (map{"fmt","xtrans","trans8"}: imm's is not in the reference tree), so this
 *              reader pins nothing about imm's bytes -- "parity unpinned by the reference" stays.
 *
 * Written separately from the product's readers (deciphon-old_amd/csrc/dcp_model.cpp,
 * deciphon-old_amd/host/dcp_lip.c): a buffer + cursor MessagePack walker that looks keys up by NAME
 * in any order, and a line/strtok text parser.  Only tests/ may link or load this.
 */
#include "oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define IO_NEG_INF (-(ofloat)INFINITY)

/* ============================================================================================== */
/* HMMER3 ASCII                                                                                     */
/* ============================================================================================== */
struct orc_h3
{
    FILE *fp;
    int entry_dist;
    ofloat eps;
    char err[160];
    unsigned line_no;
    char acc[64], name[64];
    char *cons;
};

static char const h3_amino[] = "ACDEFGHIKLMNPQRSTVWY"; /* imm_amino_iupac order = HMMER's */

/* src/model/protein_h3reader.c:79-103 */
void orc_swissprot_null(ofloat out[20])
{
    static double const f[20] = {0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198,
                                 0.0590092, 0.0594422, 0.0963728, 0.0237718, 0.0414386, 0.0482904, 0.0395639,
                                 0.0540978, 0.0683364, 0.0540687, 0.0673417, 0.0114135, 0.0304133};
    for (int i = 0; i < 20; ++i)
#ifdef ORC_F64
        out[i] = log(f[i]);
#else
        out[i] = logf((float)f[i]); /* imm_log on an imm_float literal */
#endif
}

struct orc_h3 *orc_h3_open(char const *path, int entry_dist, ofloat eps)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return NULL;
    struct orc_h3 *r = calloc(1, sizeof *r);
    r->fp = fp, r->entry_dist = entry_dist, r->eps = eps;
    return r;
}

void orc_h3_close(struct orc_h3 *r)
{
    if (!r) return;
    fclose(r->fp);
    free(r->cons);
    free(r);
}

char const *orc_h3_error(struct orc_h3 const *r) { return r->err; }
char const *orc_h3_acc(struct orc_h3 const *r) { return r->acc[0] ? r->acc : r->name; }
char const *orc_h3_consensus(struct orc_h3 const *r) { return r->cons ? r->cons : ""; }

/* next non-blank line, right-trimmed; grows the buffer as needed */
static int h3_line(struct orc_h3 *r, char **buf, size_t *cap)
{
    for (;;)
    {
        size_t n = 0;
        int c;
        while ((c = fgetc(r->fp)) != EOF && c != '\n')
        {
            if (n + 2 > *cap) *buf = realloc(*buf, *cap = *cap ? *cap * 2 : 256);
            (*buf)[n++] = (char)c;
        }
        if (c == EOF && n == 0) return 0;
        ++r->line_no;
        if (!*buf) *buf = malloc(*cap = 256);
        while (n && isspace((unsigned char)(*buf)[n - 1]))
            --n;
        (*buf)[n] = 0;
        if (n) return 1;
        if (c == EOF) return 0;
    }
}

/* -ln p -> ln p; '*' -> -inf.  Through double, then one cast (hmr holds doubles, the reader casts). */
static int h3_value(char const *tok, ofloat *out)
{
    if (!strcmp(tok, "*"))
    {
        *out = IO_NEG_INF;
        return 1;
    }
    char *end;
    double v = strtod(tok, &end);
    if (end == tok || *end || !(v >= 0)) return 0;
    *out = (ofloat)(0.0 - v); /* 0 - 0 = +0: no negative zero */
    return 1;
}

static int h3_fail(struct orc_h3 *r, char const *what)
{
    snprintf(r->err, sizeof r->err, "line %u: %s", r->line_no, what);
    return ORC_EFAIL;
}

/* `want` whitespace-separated numeric fields starting at field `skip` of the line */
static int h3_fields(char *line, unsigned skip, unsigned want, ofloat *out, char **rest_fields, unsigned *nrest)
{
    unsigned i = 0, got = 0, nr = 0;
    char *save = NULL;
    for (char *tok = strtok_r(line, " \t", &save); tok; tok = strtok_r(NULL, " \t", &save), ++i)
    {
        if (i < skip) continue;
        if (got < want)
        {
            if (!h3_value(tok, &out[got])) return 0;
            ++got;
        }
        else if (rest_fields && nr < 8) rest_fields[nr++] = tok;
    }
    if (nrest) *nrest = nr;
    return got == want;
}

static unsigned count_fields(char const *line)
{
    unsigned n = 0;
    int in = 0;
    for (; *line; ++line)
    {
        int sp = isspace((unsigned char)*line);
        if (!sp && !in) ++n;
        in = !sp;
    }
    return n;
}

/* ORC_OK + *out, ORC_END at end of file, ORC_EFAIL (message in orc_h3_error) on malformed input,
 * ORC_EINVAL for a core size outside 1..4096 (protein_model_setup, protein_model.c:157-160). */
int orc_h3_next(struct orc_h3 *r, struct orc_profile **out)
{
    *out = NULL;
    char *ln = NULL;
    size_t cap = 0;
    int rc = ORC_OK;
    ofloat *match = NULL, *trans = NULL;
    if (!h3_line(r, &ln, &cap))
    {
        free(ln);
        return ORC_END;
    }
    if (strncmp(ln, "HMMER3", 6))
    {
        rc = h3_fail(r, "not a HMMER3 profile");
        goto done;
    }
    long leng = -1;
    int amino = 0;
    r->acc[0] = r->name[0] = 0;
    for (;;)
    {
        if (!h3_line(r, &ln, &cap))
        {
            rc = h3_fail(r, "file ends inside the header");
            goto done;
        }
        char key[16] = {0}, val[64] = {0};
        int nf = sscanf(ln, "%15s %63s", key, val);
        if (nf >= 1 && !strcmp(key, "HMM")) break;
        if (nf < 2) continue;
        if (!strcmp(key, "NAME")) snprintf(r->name, sizeof r->name, "%s", val);
        else if (!strcmp(key, "ACC")) snprintf(r->acc, sizeof r->acc, "%s", val);
        else if (!strcmp(key, "LENG")) leng = strtol(val, NULL, 10);
        else if (!strcmp(key, "ALPH")) amino = !strcmp(val, "amino");
    }
    if (!amino)
    {
        rc = h3_fail(r, "ALPH is not amino");
        goto done;
    }
    { /* "HMM  A C D ... Y": the column order the values are read in */
        char *save = NULL, *tok = strtok_r(ln, " \t", &save);
        for (int i = 0; i < 20; ++i)
        {
            tok = strtok_r(NULL, " \t", &save);
            if (!tok || tok[0] != h3_amino[i] || tok[1])
            {
                rc = h3_fail(r, "amino columns are not ACDEFGHIKLMNPQRSTVWY");
                goto done;
            }
        }
    }
    if (!h3_line(r, &ln, &cap)) /* m->m m->i ... */
    {
        rc = h3_fail(r, "no transition header");
        goto done;
    }
    if (leng < 1 || leng > ORC_CORE_SIZE_MAX)
    {
        snprintf(r->err, sizeof r->err, "core size %ld out of range", leng);
        rc = ORC_EINVAL;
        goto done;
    }
    unsigned const M = (unsigned)leng;
    match = malloc(sizeof(ofloat) * 20 * M);
    trans = malloc(sizeof(ofloat) * 7 * (M + 1));
    free(r->cons);
    r->cons = calloc(M + 1, 1);
    memset(r->cons, '-', M);

    /* begin node: optional COMPO, insert emissions (ignored as the reference ignores them), transitions */
    if (!h3_line(r, &ln, &cap)) goto eof;
    if (!strncmp(ln + strspn(ln, " \t"), "COMPO", 5) && !h3_line(r, &ln, &cap)) goto eof;
    if (count_fields(ln) != 20)
    {
        rc = h3_fail(r, "expected node 0's insert emissions");
        goto done;
    }
    if (!h3_line(r, &ln, &cap)) goto eof;
    if (count_fields(ln) != 7 || !h3_fields(ln, 0, 7, trans, NULL, NULL))
    {
        rc = h3_fail(r, "bad transitions of node 0");
        goto done;
    }
    for (unsigned k = 1; k <= M; ++k)
    {
        if (!h3_line(r, &ln, &cap)) goto eof;
        if (strtol(ln, NULL, 10) != (long)k)
        {
            rc = h3_fail(r, "node index out of sequence");
            goto done;
        }
        char *rest[8];
        unsigned nrest = 0;
        if (!h3_fields(ln, 1, 20, match + 20 * (k - 1), rest, &nrest))
        {
            rc = h3_fail(r, "bad match emissions");
            goto done;
        }
        if (nrest >= 2 && rest[1][0] && !rest[1][1]) r->cons[k - 1] = rest[1][0]; /* MAP CONS RF MM CS */
        if (!h3_line(r, &ln, &cap)) goto eof;
        if (count_fields(ln) != 20)
        {
            rc = h3_fail(r, "bad insert line");
            goto done;
        }
        if (!h3_line(r, &ln, &cap)) goto eof;
        if (count_fields(ln) != 7 || !h3_fields(ln, 0, 7, trans + 7 * k, NULL, NULL))
        {
            rc = h3_fail(r, "bad transition line");
            goto done;
        }
    }
    if (!h3_line(r, &ln, &cap) || strcmp(ln, "//"))
    {
        rc = h3_fail(r, "missing // terminator");
        goto done;
    }
    ofloat null_lp[20];
    orc_swissprot_null(null_lp);
    *out = orc_profile_new(M, r->entry_dist, r->eps, null_lp, match, trans);
    if (!*out) rc = ORC_EFAIL;
    goto done;
eof:
    rc = h3_fail(r, "file ends inside the model");
done:
    free(ln);
    free(match);
    free(trans);
    return rc;
}

/* ============================================================================================== */
/* MessagePack ".dcp"                                                                               */
/* ============================================================================================== */
enum mp_kind
{
    MP_NIL,
    MP_BOOL,
    MP_UINT,
    MP_INT,
    MP_FLOAT,
    MP_STR,
    MP_BIN,
    MP_ARRAY,
    MP_MAP,
    MP_EXT,
    MP_BAD
};

struct mp_val
{
    enum mp_kind kind;
    uint64_t u;           /* UINT / BOOL; INT as two's complement */
    double f;             /* FLOAT */
    uint8_t const *data;  /* STR / BIN / EXT payload; ARRAY / MAP: first child */
    uint64_t len;         /* payload bytes; ARRAY: items; MAP: pairs */
    int ext_type;         /* EXT */
    uint8_t const *next;  /* STR / BIN / EXT / scalars: byte after this value (containers: see mp_skip) */
};

static uint64_t be(uint8_t const *p, unsigned n)
{
    uint64_t v = 0;
    for (unsigned i = 0; i < n; ++i)
        v = v << 8 | p[i];
    return v;
}

/* decode the value at p (p < end); containers are not descended into */
static struct mp_val mp_read(uint8_t const *p, uint8_t const *end)
{
    struct mp_val v = {MP_BAD, 0, 0, NULL, 0, 0, NULL};
    if (p >= end) return v;
    uint8_t const t = *p++;
#define NEED(n)                                                                                                    \
    if ((uint64_t)(end - p) < (uint64_t)(n)) return v
    if (t <= 0x7f) v.kind = MP_UINT, v.u = t, v.next = p;
    else if (t >= 0xe0) v.kind = MP_INT, v.u = (uint64_t)(int64_t)(int8_t)t, v.next = p;
    else if (t >= 0xa0 && t <= 0xbf)
    {
        NEED(t & 31);
        v.kind = MP_STR, v.data = p, v.len = t & 31, v.next = p + v.len;
    }
    else if (t >= 0x90 && t <= 0x9f) v.kind = MP_ARRAY, v.len = t & 15, v.data = p;
    else if (t >= 0x80 && t <= 0x8f) v.kind = MP_MAP, v.len = t & 15, v.data = p;
    else
        switch (t)
        {
        case 0xc0: v.kind = MP_NIL, v.next = p; break;
        case 0xc2:
        case 0xc3: v.kind = MP_BOOL, v.u = t & 1, v.next = p; break;
        case 0xca:
        {
            NEED(4);
            uint32_t b = (uint32_t)be(p, 4);
            float f;
            memcpy(&f, &b, 4);
            v.kind = MP_FLOAT, v.f = f, v.next = p + 4;
            break;
        }
        case 0xcb:
        {
            NEED(8);
            uint64_t b = be(p, 8);
            memcpy(&v.f, &b, 8);
            v.kind = MP_FLOAT, v.next = p + 8;
            break;
        }
        case 0xcc:
        case 0xcd:
        case 0xce:
        case 0xcf:
        {
            unsigned n = 1u << (t - 0xcc);
            NEED(n);
            v.kind = MP_UINT, v.u = be(p, n), v.next = p + n;
            break;
        }
        case 0xd0:
        case 0xd1:
        case 0xd2:
        case 0xd3:
        {
            unsigned n = 1u << (t - 0xd0);
            NEED(n);
            uint64_t raw = be(p, n);
            if (n < 8 && (raw >> (8 * n - 1))) raw |= ~0ull << (8 * n); /* sign-extend */
            v.kind = MP_INT, v.u = raw, v.next = p + n;
            break;
        }
        case 0xd9:
        case 0xda:
        case 0xdb:
        case 0xc4:
        case 0xc5:
        case 0xc6:
        {
            unsigned n = (t == 0xd9 || t == 0xc4) ? 1 : (t == 0xda || t == 0xc5) ? 2 : 4;
            NEED(n);
            uint64_t len = be(p, n);
            p += n;
            NEED(len);
            v.kind = t >= 0xd9 ? MP_STR : MP_BIN, v.data = p, v.len = len, v.next = p + len;
            break;
        }
        case 0xdc:
        case 0xdd:
        case 0xde:
        case 0xdf:
        {
            unsigned n = (t & 1) ? 4 : 2;
            NEED(n);
            v.kind = t <= 0xdd ? MP_ARRAY : MP_MAP, v.len = be(p, n), v.data = p + n;
            break;
        }
        case 0xd4:
        case 0xd5:
        case 0xd6:
        case 0xd7:
        case 0xd8:
        {
            uint64_t len = 1ull << (t - 0xd4);
            NEED(1 + len);
            v.kind = MP_EXT, v.ext_type = (int8_t)p[0], v.data = p + 1, v.len = len, v.next = p + 1 + len;
            break;
        }
        case 0xc7:
        case 0xc8:
        case 0xc9:
        {
            unsigned n = 1u << (t - 0xc7);
            NEED(n + 1);
            uint64_t len = be(p, n);
            p += n;
            NEED(1 + len);
            v.kind = MP_EXT, v.ext_type = (int8_t)p[0], v.data = p + 1, v.len = len, v.next = p + 1 + len;
            break;
        }
        default: break;
        }
#undef NEED
    return v;
}

/* first byte after the whole value at p (descends into containers); NULL on malformed input */
static uint8_t const *mp_skip(uint8_t const *p, uint8_t const *end)
{
    struct mp_val v = mp_read(p, end);
    if (v.kind == MP_BAD) return NULL;
    if (v.kind != MP_ARRAY && v.kind != MP_MAP) return v.next;
    uint64_t n = v.kind == MP_MAP ? 2 * v.len : v.len;
    p = v.data;
    for (uint64_t i = 0; i < n && p; ++i)
        p = mp_skip(p, end);
    return p;
}

/* value stored under `key` in the map at `map` (keys compared as strings, any order); NULL if absent */
static uint8_t const *mp_map_find(uint8_t const *map, uint8_t const *end, char const *key)
{
    struct mp_val m = mp_read(map, end);
    if (m.kind != MP_MAP) return NULL;
    uint8_t const *p = m.data;
    size_t const kl = strlen(key);
    for (uint64_t i = 0; i < m.len; ++i)
    {
        struct mp_val k = mp_read(p, end);
        uint8_t const *val = mp_skip(p, end);
        if (!val) return NULL;
        if (k.kind == MP_STR && k.len == kl && !memcmp(k.data, key, kl)) return val;
        p = mp_skip(val, end);
        if (!p) return NULL;
    }
    return NULL;
}

/* lite-pack "1darray": an ext object whose type byte names the element type and whose payload is the
 * big-endian elements.  lite-pack itself is absent from the reference tree, so the type byte's VALUE is
 * the product's choice and is not checked here: the payload length fixes the element count. */
static int mp_f32_array(uint8_t const *p, uint8_t const *end, unsigned want, ofloat *out)
{
    struct mp_val v = mp_read(p, end);
    if (v.kind != MP_EXT || v.len != 4ull * want) return 0;
    for (unsigned i = 0; i < want; ++i)
    {
        uint32_t b = (uint32_t)be(v.data + 4 * i, 4);
        float f;
        memcpy(&f, &b, 4);
        out[i] = f;
    }
    return 1;
}

struct orc_dcp
{
    uint8_t *buf;
    size_t size;
    uint8_t const **prof; /* start of every profile's map */
    unsigned nprofiles;
    int entry_dist;
    ofloat epsilon;
    unsigned float_size, magic, typeid_;
    char err[160];
};

static struct orc_dcp *dcp_fail(struct orc_dcp *d, char *err, size_t cap, char const *what)
{
    if (err && cap) snprintf(err, cap, "%s", what);
    if (d)
    {
        free(d->buf);
        free(d->prof);
        free(d);
    }
    return NULL;
}

struct orc_dcp *orc_dcp_open(char const *path, char *err, size_t errcap)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return dcp_fail(NULL, err, errcap, "cannot open");
    struct orc_dcp *d = calloc(1, sizeof *d);
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    d->buf = malloc(sz > 0 ? (size_t)sz : 1);
    d->size = sz > 0 ? (size_t)sz : 0;
    if (fread(d->buf, 1, d->size, fp) != d->size)
    {
        fclose(fp);
        return dcp_fail(d, err, errcap, "short read");
    }
    fclose(fp);
    uint8_t const *end = d->buf + d->size;
    uint8_t const *hdr = mp_map_find(d->buf, end, "header");
    uint8_t const *profs = mp_map_find(d->buf, end, "profiles");
    if (!hdr || !profs) return dcp_fail(d, err, errcap, "root map lacks header / profiles");
    struct mp_val v;
#define FIELD(name) mp_read(mp_map_find(hdr, end, name) ? mp_map_find(hdr, end, name) : end, end)
    v = FIELD("magic_number");
    if (v.kind != MP_UINT || v.u != 0xC6F0) return dcp_fail(d, err, errcap, "bad magic number"); /* db/types.h:11 */
    d->magic = (unsigned)v.u;
    v = FIELD("profile_typeid");
    if (v.kind != MP_UINT) return dcp_fail(d, err, errcap, "bad profile_typeid");
    d->typeid_ = (unsigned)v.u;
    v = FIELD("float_size");
    if (v.kind != MP_UINT || v.u != 4) return dcp_fail(d, err, errcap, "float_size is not 4");
    d->float_size = 4;
    v = FIELD("entry_dist");
    if (v.kind != MP_UINT || v.u < 1 || v.u > 2) return dcp_fail(d, err, errcap, "bad entry_dist");
    d->entry_dist = (int)v.u;
    v = FIELD("epsilon");
    if (v.kind != MP_FLOAT || !(v.f >= 0 && v.f <= 1)) return dcp_fail(d, err, errcap, "bad epsilon");
    d->epsilon = (ofloat)v.f;
    v = FIELD("profile_sizes");
    if (v.kind != MP_EXT || v.len % 4) return dcp_fail(d, err, errcap, "bad profile_sizes");
#undef FIELD
    unsigned const n_sizes = (unsigned)(v.len / 4);
    uint8_t const *sizes = v.data;
    struct mp_val arr = mp_read(profs, end);
    if (arr.kind != MP_ARRAY || arr.len != n_sizes) return dcp_fail(d, err, errcap, "profiles array does not match profile_sizes");
    d->nprofiles = n_sizes;
    d->prof = malloc(sizeof *d->prof * (n_sizes ? n_sizes : 1));
    uint8_t const *p = arr.data;
    for (unsigned i = 0; i < n_sizes; ++i)
    {
        d->prof[i] = p;
        uint8_t const *q = mp_skip(p, end);
        /* profile_sizes[i] = bytes of profile i: what profile_reader's partition offsets are summed from */
        if (!q || (uint64_t)(q - p) != be(sizes + 4 * i, 4)) return dcp_fail(d, err, errcap, "profile size mismatch");
        p = q;
    }
    if (p != end) return dcp_fail(d, err, errcap, "bytes after the last profile");
    return d;
}

void orc_dcp_close(struct orc_dcp *d) { dcp_fail(d, NULL, 0, ""); }
unsigned orc_dcp_nprofiles(struct orc_dcp const *d) { return d->nprofiles; }
int orc_dcp_entry_dist(struct orc_dcp const *d) { return d->entry_dist; }
ofloat orc_dcp_epsilon(struct orc_dcp const *d) { return d->epsilon; }

static int read_ndist(uint8_t const *p, uint8_t const *end, struct orc_nuclt_dist *out)
{
    struct mp_val a = mp_read(p, end);
    if (a.kind != MP_ARRAY || a.len != 2) return 0;
    uint8_t const *second = mp_skip(a.data, end);
    return second && mp_f32_array(a.data, end, 4, out->nucltp) && mp_f32_array(second, end, 125, out->codonm);
}

/* Profile i: core size (always), and whichever of the outputs are not NULL:
 * accession [32]; trans8 [8][M] rows entry, MM, IM, DM, MD, DD, MI, II; xtrans [13] as the alt dp was packed
 * (order RR.. of orc_xtrans; RR from the null dp); null / insert dists; match dists [M]; consensus [M+1].
 * ORC_OK, or ORC_EFAIL for a layout this reader does not know (e.g. imm's own dp encoding). */
int orc_dcp_profile(struct orc_dcp const *d, unsigned i, unsigned *core_size, char *acc, ofloat *trans8,
                    ofloat *xtrans, struct orc_nuclt_dist *null_d, struct orc_nuclt_dist *insert_d,
                    struct orc_nuclt_dist *match_d, char *consensus)
{
    if (i >= d->nprofiles) return ORC_EINVAL;
    uint8_t const *end = d->buf + d->size, *m = d->prof[i];
    struct mp_val v = mp_read(m, end);
    if (v.kind != MP_MAP || v.len != 16) return ORC_EFAIL; /* protein_profile.c:43-44 */
    uint8_t const *p;
    if (!(p = mp_map_find(m, end, "core_size"))) return ORC_EFAIL;
    v = mp_read(p, end);
    if (v.kind != MP_UINT || v.u < 1 || v.u > ORC_CORE_SIZE_MAX) return ORC_EFAIL;
    unsigned const M = (unsigned)v.u;
    *core_size = M;
    if (acc)
    {
        if (!(p = mp_map_find(m, end, "accession"))) return ORC_EFAIL;
        v = mp_read(p, end);
        if (v.kind != MP_STR || v.len >= 32) return ORC_EFAIL;
        memcpy(acc, v.data, v.len);
        acc[v.len] = 0;
    }
    if (consensus)
    {
        if (!(p = mp_map_find(m, end, "consensus"))) return ORC_EFAIL;
        v = mp_read(p, end);
        if (v.kind != MP_STR || v.len != M) return ORC_EFAIL;
        memcpy(consensus, v.data, M);
        consensus[M] = 0;
    }
    if (trans8 || xtrans)
    {
        uint8_t const *alt = mp_map_find(m, end, "alt"), *nul = mp_map_find(m, end, "null");
        if (!alt || !nul) return ORC_EFAIL;
        struct mp_val fmt = mp_read(mp_map_find(alt, end, "fmt") ? mp_map_find(alt, end, "fmt") : end, end);
        if (fmt.kind != MP_STR || fmt.len != 8 || memcmp(fmt.data, "dcp-dp-1", 8)) return ORC_EFAIL;
        if (trans8 && (!(p = mp_map_find(alt, end, "trans8")) || !mp_f32_array(p, end, 8 * M, trans8))) return ORC_EFAIL;
        if (xtrans)
        {
            ofloat rr;
            if (!(p = mp_map_find(alt, end, "xtrans")) || !mp_f32_array(p, end, 13, xtrans)) return ORC_EFAIL;
            if (!(p = mp_map_find(nul, end, "xtrans")) || !mp_f32_array(p, end, 1, &rr)) return ORC_EFAIL;
            xtrans[0] = rr;
        }
    }
    if (null_d && (!(p = mp_map_find(m, end, "null_ndist")) || !read_ndist(p, end, null_d))) return ORC_EFAIL;
    if (insert_d && (!(p = mp_map_find(m, end, "alt_insert_ndist")) || !read_ndist(p, end, insert_d))) return ORC_EFAIL;
    if (match_d)
    {
        if (!(p = mp_map_find(m, end, "alt_match_ndist"))) return ORC_EFAIL;
        v = mp_read(p, end);
        if (v.kind != MP_ARRAY || v.len != M) return ORC_EFAIL;
        p = v.data;
        for (unsigned k = 0; k < M; ++k)
        {
            if (!read_ndist(p, end, match_d + k)) return ORC_EFAIL;
            p = mp_skip(p, end);
        }
    }
    return ORC_OK;
}

/* (null, alt) score of `seq` against profile i of the file, from the file's bytes alone: emission tables
 * by orc_frame_table from the stored nuclt_dists, the stored core transitions, and the 13 special
 * transitions of protein_profile_setup for this length (what thread_run sets before each pair). */
int orc_dcp_score(struct orc_dcp const *d, unsigned i, unsigned char const *seq, unsigned L, int multi_hits,
                  int hmmer3_compat, ofloat *null_loglik, ofloat *alt_loglik)
{
    unsigned M = 0;
    int rc = orc_dcp_profile(d, i, &M, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    if (rc) return rc;
    ofloat *trans8 = malloc(sizeof(ofloat) * 8 * M);
    struct orc_nuclt_dist nd, id, *md = malloc(sizeof *md * M);
    ofloat *em = malloc(sizeof(ofloat) * ORC_NCODES * (size_t)M), *row = malloc(sizeof(ofloat) * ORC_NCODES);
    ofloat ei[ORC_NCODES], en[ORC_NCODES], xt[13];
    rc = orc_dcp_profile(d, i, &M, NULL, trans8, NULL, &nd, &id, md, NULL);
    if (!rc) rc = orc_xtrans(L, multi_hits, hmmer3_compat, xt);
    if (!rc)
    {
        orc_frame_table(&nd, d->epsilon, en);
        orc_frame_table(&id, d->epsilon, ei);
        for (unsigned k = 0; k < M; ++k)
        {
            orc_frame_table(&md[k], d->epsilon, row);
            for (unsigned c = 0; c < ORC_NCODES; ++c)
                em[(size_t)c * M + k] = row[c];
        }
        rc = orc_dp_tables(M, M, trans8, em, ei, en, xt, seq, L, null_loglik, alt_loglik);
    }
    free(trans8), free(md), free(em), free(row);
    return rc;
}
