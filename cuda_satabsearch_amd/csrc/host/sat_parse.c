/*
 * sat_parse.c - ASCII tableau + SSE-distance-matrix reader.  See sat_parse.h.
 *
 * Cell semantics kept from the reference reader (they are what existing
 * databases rely on, quirks included):
 *   - a tableau cell j of a row sits at text columns 3j, 3j+1
 *     (parsetableaux.c:216-217); the diagonal cell is an SSE type: 'e?' -> 0,
 *     otherwise by second letter a/i/g -> 1/2/3 (parsetableaux.c:52-76);
 *     off-diagonal: first letter P R O L ? -> high nibble 0..4, second letter
 *     E D S T ? -> low nibble 0..4 (parsetableaux.c:88-140); anything else is
 *     fatal (exit status 1);
 *   - a distance cell j is strtof() at text column 7j (parsetableaux.c:288), so
 *     a value >= 100 A (printed 7 wide by the db builder) shifts the rest of
 *     that row: reproduced as is;
 *   - the record header is read with fscanf("%8s %d\n") (parsetableaux.c:391),
 *     which also swallows the blank separator line and any leading blanks of
 *     the first tableau row;
 *   - order > 111 is skipped with a warning (parsetableaux.c:457-465).
 */
#include <ctype.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "sat_parse.h"

void sat_set_init(sat_struct_set *set)
{
    memset(set, 0, sizeof(*set));
}

void sat_set_free(sat_struct_set *set)
{
    free(set->order);
    free(set->name);
    free(set->cell_off);
    free(set->tab);
    free(set->dist);
    memset(set, 0, sizeof(*set));
}

static int grow_entries(sat_struct_set *set)
{
    if (set->count < set->capacity)
        return 0;
    int ncap = set->capacity ? set->capacity * 2 : 1024;
    int *o = (int *)realloc(set->order, (size_t)ncap * sizeof(int));
    if (!o) return -1;
    set->order = o;
    char *nm = (char *)realloc(set->name, (size_t)ncap * (SAT_LABELSIZE + 1));
    if (!nm) return -1;
    set->name = nm;
    int64_t *co = (int64_t *)realloc(set->cell_off, (size_t)ncap * sizeof(int64_t));
    if (!co) return -1;
    set->cell_off = co;
    set->capacity = ncap;
    return 0;
}

static int grow_cells(sat_struct_set *set, int64_t extra)
{
    if (set->cells + extra <= set->cells_cap)
        return 0;
    int64_t ncap = set->cells_cap ? set->cells_cap * 2 : (1 << 16);
    while (ncap < set->cells + extra)
        ncap *= 2;
    uint8_t *t = (uint8_t *)realloc(set->tab, (size_t)ncap);
    if (!t) return -1;
    set->tab = t;
    float *d = (float *)realloc(set->dist, (size_t)ncap * sizeof(float));
    if (!d) return -1;
    set->dist = d;
    set->cells_cap = ncap;
    return 0;
}

/* room for at least `count` entries and `cells` cells in total */
static int reserve(sat_struct_set *set, int count, int64_t cells)
{
    if (count > set->capacity) {
        int *o = (int *)realloc(set->order, (size_t)count * sizeof(int));
        if (!o) return -1;
        set->order = o;
        char *nm = (char *)realloc(set->name, (size_t)count * (SAT_LABELSIZE + 1));
        if (!nm) return -1;
        set->name = nm;
        int64_t *co = (int64_t *)realloc(set->cell_off, (size_t)count * sizeof(int64_t));
        if (!co) return -1;
        set->cell_off = co;
        set->capacity = count;
    }
    if (cells > set->cells_cap) {
        uint8_t *t = (uint8_t *)realloc(set->tab, (size_t)cells);
        if (!t) return -1;
        set->tab = t;
        float *d = (float *)realloc(set->dist, (size_t)cells * sizeof(float));
        if (!d) return -1;
        set->dist = d;
        set->cells_cap = cells;
    }
    return 0;
}

int sat_set_append(sat_struct_set *set, const char *name, int order,
                   const uint8_t *tab_tri, const float *dist_tri)
{
    int64_t ncell = order > 0 ? (int64_t)order * (order + 1) / 2 : 0;
    if (grow_entries(set) || grow_cells(set, ncell))
        return -1;
    int s = set->count++;
    set->order[s] = order;
    char *dst = set->name + (size_t)s * (SAT_LABELSIZE + 1);
    memset(dst, 0, SAT_LABELSIZE + 1);
    strncpy(dst, name, SAT_LABELSIZE);
    set->cell_off[s] = set->cells;
    if (ncell) {
        memcpy(set->tab + set->cells, tab_tri, (size_t)ncell);
        memcpy(set->dist + set->cells, dist_tri, (size_t)ncell * sizeof(float));
    }
    set->cells += ncell;
    return s;
}

/* A piece of the threaded reader that was cut at a false record header must not end the process on the
 * "bad code" it then meets (the sequential parse, which defines the result, may read the same bytes as
 * distances): a piece parser points this at its own flag, a bad code sets it and the piece is thrown away. */
static __thread int *t_soft_error = NULL;

static uint8_t ssetype_code(const char *c)
{
    if (c[0] == 'e')
        return SAT_SSE_E;
    switch (c[1]) {
    case 'a': return SAT_SSE_XA;
    case 'i': return SAT_SSE_XI;
    case 'g': return SAT_SSE_XG;
    default:
        if (t_soft_error) { *t_soft_error = 1; return 0; }
        fprintf(stderr, "Bad helix type %c\n", c[1]);
        exit(1);
    }
}

static uint8_t nibble_of(char c, const char *alphabet)
{
    const char *p = c ? strchr(alphabet, c) : NULL;
    if (!p) {
        if (t_soft_error) { *t_soft_error = 1; return 0; }
        fprintf(stderr, "invalid tableaux code %c\n", c);
        exit(1);
    }
    return (uint8_t)(p - alphabet);
}

static uint8_t tableau_code(const char *c)
{
    /* position in the alphabet string is the nibble value; '?' is 4 */
    uint8_t hi = nibble_of(c[0], "PROL?");
    uint8_t lo = nibble_of(c[1], "EDST?");
    return (uint8_t)((hi << 4) | lo);
}

static void read_line(FILE *fp, char *buf)
{
    memset(buf, 0, SAT_MAX_LINE_LEN);
    if (!fgets(buf, SAT_MAX_LINE_LEN, fp))
        buf[0] = '\0';
}

int sat_read_structures(FILE *fp, sat_struct_set *set, const char *what)
{
    char buf[SAT_MAX_LINE_LEN];            /* automatic: the readers may run on several threads */
    uint8_t *tri_tab = (uint8_t *)malloc(SAT_MAXDIM * (SAT_MAXDIM + 1) / 2);
    float *tri_dist = (float *)malloc(SAT_MAXDIM * (SAT_MAXDIM + 1) / 2 * sizeof(float));
    char name[SAT_LABELSIZE + 1];
    int order, added = 0, skipped = 0;

    if (!tri_tab || !tri_dist) {
        free(tri_tab);
        free(tri_dist);
        return -1;
    }
    while (!feof(fp)) {
        if (fscanf(fp, "%8s %d\n", name, &order) != 2)
            break; /* end of input */
        if (order > SAT_MAXDIM) {
            fprintf(stderr, "Tableau %s order %d is too large (max is %d)\n",
                    name, order, SAT_MAXDIM);
            fprintf(stderr, "WARNING: excluded %s structure %s as it is too large\n",
                    what, name);
            for (int i = 0; i < 2 * order; i++)
                read_line(fp, buf);
            skipped++;
            continue;
        }
        if (order < 1) {
            /* no rows follow a record of order 0; a negative order has no meaning (the reference
             * keeps such a record and then indexes with it: undefined there, dropped here) */
            fprintf(stderr, "WARNING: excluded %s structure %s: order %d is not positive\n", what, name, order);
            set->skipped++;
            continue;
        }
        int64_t c = 0;
        for (int i = 0; i < order; i++) {
            read_line(fp, buf);
            for (int j = 0; j <= i; j++, c++)
                tri_tab[c] = (i == j) ? ssetype_code(&buf[3 * j])
                                      : tableau_code(&buf[3 * j]);
        }
        c = 0;
        for (int i = 0; i < order; i++) {
            read_line(fp, buf);
            for (int j = 0; j <= i; j++, c++)
                tri_dist[c] = strtof(&buf[7 * j], NULL);
        }
        if (sat_set_append(set, name, order, tri_tab, tri_dist) < 0) {
            free(tri_tab);
            free(tri_dist);
            return -1;
        }
        added++;
    }
    if (skipped > 0)
        fprintf(stderr, "WARNING: skipped %d %s tableaux of order > %d\n",
                skipped, what, SAT_MAXDIM);
    set->skipped += skipped;
    free(tri_tab);
    free(tri_dist);
    return added;
}

void sat_set_expand(const sat_struct_set *set, int s, int pitch,
                    uint8_t *tab_dense, float *dist_dense)
{
    int n = set->order[s];
    const uint8_t *t = set->tab + set->cell_off[s];
    const float *d = set->dist + set->cell_off[s];
    int64_t c = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++, c++) {
            tab_dense[(size_t)i * pitch + j] = t[c];
            tab_dense[(size_t)j * pitch + i] = t[c];
            dist_dense[(size_t)i * pitch + j] = d[c];
            dist_dense[(size_t)j * pitch + i] = d[c];
        }
}

/* ------------------------------------------------------------------ memory-image reader */

typedef struct cursor { const char *p, *end; } cursor;

static int is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

static void skip_space(cursor *c)
{
    while (c->p < c->end && is_space(*c->p)) c->p++;
}

/* fscanf("%8s %d\n"): skip blanks, up to 8 non-blank chars, blanks, a decimal int, blanks */
static int scan_header(cursor *c, char *name, int *order)
{
    skip_space(c);
    int n = 0;
    while (c->p < c->end && !is_space(*c->p) && n < SAT_LABELSIZE) name[n++] = *c->p++;
    name[n] = '\0';
    if (n == 0) return 0;
    skip_space(c);
    const char *q = c->p;
    int neg = 0;
    if (q < c->end && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
    if (q >= c->end || !isdigit((unsigned char)*q)) return 1;        /* name read, order missing */
    long v = 0;
    while (q < c->end && isdigit((unsigned char)*q)) { v = v * 10 + (*q - '0'); if (v > 100000000) v = 100000000; q++; }
    *order = (int)(neg ? -v : v);
    c->p = q;
    skip_space(c);
    return 2;
}

/* fgets(): one line (without the newline) copied into buf; everything after it is zero, as
 * in the zero-filled stdio buffer (only the part the previous line dirtied is cleared) */
static void next_line(cursor *c, char *buf, int *dirty)
{
    const char *nl = (const char *)memchr(c->p, '\n', (size_t)(c->end - c->p));
    size_t n = nl ? (size_t)(nl - c->p) : (size_t)(c->end - c->p);
    size_t take = n < SAT_MAX_LINE_LEN - 1 ? n : SAT_MAX_LINE_LEN - 1;
    memcpy(buf, c->p, take);
    if ((int)take < *dirty) memset(buf + take, 0, (size_t)*dirty - take);
    buf[take] = '\0';
    *dirty = (int)take;
    /* a longer line would be continued by the next fgets call; same here */
    c->p += take < n ? take : (nl ? n + 1 : n);
}

/* strtof(&buf[7j]) with a fast path for "[blanks][d]dd.ddd" followed by a blank or the end */
static float distance_at(const char *p)
{
    const char *q = p;
    while (*q == ' ') q++;
    unsigned v = 0;
    int nd = 0;
    while (*q >= '0' && *q <= '9' && nd < 4) { v = v * 10 + (unsigned)(*q - '0'); q++; nd++; }
    if (nd >= 1 && nd <= 3 && *q == '.' &&
        q[1] >= '0' && q[1] <= '9' && q[2] >= '0' && q[2] <= '9' && q[3] >= '0' && q[3] <= '9' &&
        (q[4] == ' ' || q[4] == '\0')) {
        v = v * 1000 + (unsigned)(q[1] - '0') * 100 + (unsigned)(q[2] - '0') * 10 + (unsigned)(q[3] - '0');
        /* == strtof() of the same text for every v < 10^6 (checked exhaustively in tests) */
        return (float)((double)v / 1000.0);
    }
    return strtof(p, NULL);
}

float sat_distance_cell(const char *text)
{
    return distance_at(text);
}

/* `incomplete` (may be NULL; given by the piece parsers of the threaded reader): set when the span ended inside a
 * record - rows still missing at its end - or a cell held no valid code; warnings are then left to the
 * sequential parse that follows. */
static int read_structures_span(const char *text, size_t len, sat_struct_set *set, const char *what, size_t *consumed,
                                int *incomplete)
{
    char buf[SAT_MAX_LINE_LEN];
    uint8_t *tri_tab = (uint8_t *)malloc(SAT_MAXDIM * (SAT_MAXDIM + 1) / 2);
    float *tri_dist = (float *)malloc(SAT_MAXDIM * (SAT_MAXDIM + 1) / 2 * sizeof(float));
    char name[SAT_LABELSIZE + 1];
    int order = 0, added = 0, skipped = 0;
    cursor c = { text, text + len };
    int dirty = SAT_MAX_LINE_LEN - 1;
    memset(buf, 0, SAT_MAX_LINE_LEN);

    if (!tri_tab || !tri_dist) {
        free(tri_tab);
        free(tri_dist);
        return -1;
    }
    int cut_short = 0;
    t_soft_error = incomplete ? &cut_short : NULL;
    while (c.p < c.end && !cut_short) {
        if (scan_header(&c, name, &order) != 2)
            break;
        if (order > SAT_MAXDIM) {
            fprintf(stderr, "Tableau %s order %d is too large (max is %d)\n", name, order, SAT_MAXDIM);
            fprintf(stderr, "WARNING: excluded %s structure %s as it is too large\n", what, name);
            for (int i = 0; i < 2 * order; i++) {
                if (c.p >= c.end) cut_short = 1;
                next_line(&c, buf, &dirty);
            }
            skipped++;
            continue;
        }
        if (order < 1) {
            fprintf(stderr, "WARNING: excluded %s structure %s: order %d is not positive\n", what, name, order);
            set->skipped++;
            continue;
        }
        int64_t k = 0;
        for (int i = 0; i < order; i++) {
            if (c.p >= c.end) cut_short = 1;         /* the span ends inside this record (an empty line is read) */
            next_line(&c, buf, &dirty);
            for (int j = 0; j <= i; j++, k++)
                tri_tab[k] = (i == j) ? ssetype_code(&buf[3 * j]) : tableau_code(&buf[3 * j]);
        }
        k = 0;
        for (int i = 0; i < order; i++) {
            if (c.p >= c.end) cut_short = 1;
            next_line(&c, buf, &dirty);
            for (int j = 0; j <= i; j++, k++)
                tri_dist[k] = distance_at(&buf[7 * j]);
        }
        if (sat_set_append(set, name, order, tri_tab, tri_dist) < 0) {
            t_soft_error = NULL;
            free(tri_tab);
            free(tri_dist);
            return -1;
        }
        added++;
    }
    t_soft_error = NULL;
    if (incomplete) *incomplete = cut_short;
    if (skipped > 0)
        fprintf(stderr, "WARNING: skipped %d %s tableaux of order > %d\n", skipped, what, SAT_MAXDIM);
    set->skipped += skipped;
    free(tri_tab);
    free(tri_dist);
    if (consumed) *consumed = (size_t)(c.p - text);
    return added;
}

int sat_read_structures_mem(const char *text, size_t len, sat_struct_set *set, const char *what)
{
    return read_structures_span(text, len, set, what, NULL, NULL);
}

/* ---- parallel parse of a memory image: the file is cut at record headers, every piece is parsed by
 * its own thread into its own set, the sets are concatenated.  A piece must end exactly where the next
 * begins (a record that claims more rows than it has would run into its neighbour): if any does not,
 * the whole image is parsed again sequentially, which is what defines the result. */

/* does a record header ("name order": two blank-separated tokens, the second all digits) start at p?  The line
 * before it must be blank, as the database builder writes it between records (scripts/convdb2.py:226): a row of
 * integer-formatted distances such as "12 0" has the shape of a header but follows another row.  A file without
 * blank separators simply finds no cut and is parsed sequentially. */
static int looks_like_header(const char *text, const char *p, const char *end)
{
    if (p > text) {
        const char *b = p - 1;                       /* the newline that ends the previous line */
        if (*b != '\n') return 0;
        while (b > text && b[-1] != '\n') {
            b--;
            if (*b != ' ' && *b != '\t' && *b != '\r') return 0;
        }
    }
    const char *q = p;
    while (q < end && (*q == ' ' || *q == '\t')) q++;
    int n = 0;
    while (q < end && !is_space(*q) && n <= SAT_LABELSIZE) { q++; n++; }
    if (n == 0 || n > SAT_LABELSIZE) return 0;
    if (q >= end || (*q != ' ' && *q != '\t')) return 0;
    while (q < end && (*q == ' ' || *q == '\t')) q++;
    int digits = 0;
    while (q < end && isdigit((unsigned char)*q)) { q++; digits++; }
    if (digits == 0 || digits > 4) return 0;
    while (q < end && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
    return q >= end || *q == '\n';
}

typedef struct parse_job {
    const char *text;
    size_t len;
    const char *what;
    sat_struct_set set;
    size_t consumed;
    int added;
    int incomplete;
} parse_job;

static void *parse_job_run(void *arg)
{
    parse_job *j = (parse_job *)arg;
    j->added = read_structures_span(j->text, j->len, &j->set, j->what, &j->consumed, &j->incomplete);
    return NULL;
}

int sat_read_structures_mem_mt(const char *text, size_t len, sat_struct_set *set, const char *what, int nthreads)
{
    const size_t min_piece = (size_t)1 << 20;
    if (nthreads > 64) nthreads = 64;
    if (nthreads > 1 && len / (size_t)nthreads < min_piece) nthreads = (int)(len / min_piece);
    if (nthreads < 2) return sat_read_structures_mem(text, len, set, what);

    const int count_before = set->count;
    parse_job *jobs = (parse_job *)calloc((size_t)nthreads, sizeof(parse_job));
    pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    size_t *start = (size_t *)calloc((size_t)nthreads + 1, sizeof(size_t));
    if (!jobs || !tid || !start) { free(jobs); free(tid); free(start); return -1; }
    const char *end = text + len;
    int pieces = 0;
    start[pieces++] = 0;
    for (int t = 1; t < nthreads; t++) {
        const char *p = text + len * (size_t)t / (size_t)nthreads;
        if (p <= text + start[pieces - 1]) continue;
        /* the next line start at or after p whose line looks like a record header */
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        p = nl ? nl + 1 : end;
        while (p < end && !looks_like_header(text, p, end)) {
            nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            p = nl ? nl + 1 : end;
        }
        if (p < end && (size_t)(p - text) > start[pieces - 1]) start[pieces++] = (size_t)(p - text);
    }
    start[pieces] = len;
    int ok = 1, launched = 0;
    for (int t = 0; t < pieces; t++) {
        jobs[t].text = text + start[t];
        jobs[t].len = start[t + 1] - start[t];
        jobs[t].what = what;
        sat_set_init(&jobs[t].set);
        if (pthread_create(&tid[t], NULL, parse_job_run, &jobs[t]) != 0) { ok = 0; break; }
        launched++;
    }
    for (int t = 0; t < launched; t++) pthread_join(tid[t], NULL);
    for (int t = 0; ok && t < pieces; t++) {
        if (jobs[t].added < 0 || jobs[t].incomplete) ok = 0;
        /* the piece must have been consumed to its end, blanks aside */
        const char *q = jobs[t].text + jobs[t].consumed, *pe = jobs[t].text + jobs[t].len;
        while (q < pe && is_space(*q)) q++;
        if (q != pe) ok = 0;
    }
    int total = -1;
    if (ok) {
        /* concatenate: one reservation, then whole-array copies per piece (offsets rebased) */
        int add_count = 0;
        int64_t add_cells = 0;
        for (int t = 0; t < pieces; t++) { add_count += jobs[t].set.count; add_cells += jobs[t].set.cells; }
        if (reserve(set, set->count + add_count, set->cells + add_cells) != 0) ok = 0;
        for (int t = 0; ok && t < pieces; t++) {
            const sat_struct_set *ps = &jobs[t].set;
            if (ps->count > 0) {
                memcpy(set->order + set->count, ps->order, (size_t)ps->count * sizeof(int));
                memcpy(set->name + (size_t)set->count * (SAT_LABELSIZE + 1), ps->name, (size_t)ps->count * (SAT_LABELSIZE + 1));
                for (int k = 0; k < ps->count; k++) set->cell_off[set->count + k] = ps->cell_off[k] + set->cells;
                memcpy(set->tab + set->cells, ps->tab, (size_t)ps->cells);
                memcpy(set->dist + set->cells, ps->dist, (size_t)ps->cells * sizeof(float));
                set->count += ps->count;
                set->cells += ps->cells;
            }
            set->skipped += ps->skipped;
        }
        total = ok ? add_count : -1;
    }
    for (int t = 0; t < pieces; t++) sat_set_free(&jobs[t].set);
    free(jobs); free(tid); free(start);
    if (!ok) {
        /* a piece did not end at its cut (or a thread could not be started): the sequential parse decides.
         * Nothing has been appended to `set` unless the merge itself failed - an allocation failure. */
        if (set->count != count_before) return -1;
        return sat_read_structures_mem(text, len, set, what);
    }
    return total;
}

int sat_read_structures_file(const char *path, sat_struct_set *set, const char *what)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return -1; }
    if (st.st_size == 0) { close(fd); return 0; }
    void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return -1;
    (void)madvise(map, (size_t)st.st_size, MADV_WILLNEED);
    /* big files are parsed by several threads (SAT_PARSE_THREADS overrides the count, 1 = sequential) */
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    int nthreads = cores > 16 ? 16 : (cores < 1 ? 1 : (int)cores);
    const char *ov = getenv("SAT_PARSE_THREADS");
    if (ov && atoi(ov) > 0) nthreads = atoi(ov);
    int n = sat_read_structures_mem_mt((const char *)map, (size_t)st.st_size, set, what, nthreads);
    munmap(map, (size_t)st.st_size);
    return n;
}

/* ------------------------------------------------------------------ ASCII writer */

int sat_set_write_ascii(const sat_struct_set *set, const char *path)
{
    /* the database builder's format (scripts/convdb2.py:214-226): "%-8s %4d" header, rows of
     * two-letter codes + blank, rows of "%6.3f " distances, blank line between records */
    static const char hi[] = "PROL?", lo[] = "EDST?";
    static const char *tname[4] = { "e  ", "xa ", "xi ", "xg " };
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    char *line = (char *)malloc((size_t)SAT_MAXDIM * 16 + 64);
    if (!line) { fclose(f); return -1; }
    int ok = 1;
    for (int s = 0; ok && s < set->count; s++) {
        const int n = set->order[s];
        const uint8_t *t = set->tab + set->cell_off[s];
        const float *d = set->dist + set->cell_off[s];
        ok = fprintf(f, "%-8s %4d\n", sat_set_name(set, s), n) > 0;
        int64_t c = 0;
        for (int i = 0; ok && i < n; i++) {
            char *w = line;
            for (int j = 0; j <= i; j++, c++) {
                if (i == j) {
                    memcpy(w, tname[t[c] & 3], 3);
                } else {
                    const int h = t[c] >> 4, l = t[c] & 15;
                    w[0] = hi[h < 4 ? h : 4];
                    w[1] = lo[l < 4 ? l : 4];
                    w[2] = ' ';
                }
                w += 3;
            }
            *w++ = '\n';
            ok = fwrite(line, 1, (size_t)(w - line), f) == (size_t)(w - line);
        }
        c = 0;
        for (int i = 0; ok && i < n; i++) {
            char *w = line;
            for (int j = 0; j <= i; j++, c++)
                w += snprintf(w, 16, "%6.3f ", (double)d[c]);
            *w++ = '\n';
            ok = fwrite(line, 1, (size_t)(w - line), f) == (size_t)(w - line);
        }
        ok = ok && fputc('\n', f) != EOF;
    }
    free(line);
    if (fclose(f) != 0) ok = 0;
    return ok ? 0 : -1;
}

/* ------------------------------------------------------------------ binary image */

#define SAT_BIN_MAGIC "SATBIN01"

int sat_set_save_binary(const sat_struct_set *set, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int64_t hdr[2] = { set->count, set->cells };
    int ok = fwrite(SAT_BIN_MAGIC, 1, 8, f) == 8 && fwrite(hdr, sizeof(hdr), 1, f) == 1;
    size_t n = (size_t)set->count;
    ok = ok && (n == 0 || (fwrite(set->order, sizeof(int), n, f) == n &&
                           fwrite(set->name, SAT_LABELSIZE + 1, n, f) == n &&
                           fwrite(set->cell_off, sizeof(int64_t), n, f) == n));
    ok = ok && (set->cells == 0 || (fwrite(set->tab, 1, (size_t)set->cells, f) == (size_t)set->cells &&
                                    fwrite(set->dist, sizeof(float), (size_t)set->cells, f) == (size_t)set->cells));
    if (fclose(f) != 0) ok = 0;
    return ok ? 0 : -1;
}

int sat_set_load_binary(const char *path, sat_struct_set *set)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    char magic[8];
    int64_t hdr[2];
    sat_struct_set tmp;
    sat_set_init(&tmp);
    int ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, SAT_BIN_MAGIC, 8) == 0 &&
             fread(hdr, sizeof(hdr), 1, f) == 1 && hdr[0] >= 0 && hdr[0] < (1 << 30) && hdr[1] >= 0;
    if (ok) {
        size_t n = (size_t)hdr[0], cells = (size_t)hdr[1];
        tmp.order = (int *)malloc(sizeof(int) * (n + 1));
        tmp.name = (char *)malloc((SAT_LABELSIZE + 1) * (n + 1));
        tmp.cell_off = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
        tmp.tab = (uint8_t *)malloc(cells + 1);
        tmp.dist = (float *)malloc(sizeof(float) * (cells + 1));
        ok = tmp.order && tmp.name && tmp.cell_off && tmp.tab && tmp.dist;
        ok = ok && fread(tmp.order, sizeof(int), n, f) == n && fread(tmp.name, SAT_LABELSIZE + 1, n, f) == n &&
             fread(tmp.cell_off, sizeof(int64_t), n, f) == n && fread(tmp.tab, 1, cells, f) == cells &&
             fread(tmp.dist, sizeof(float), cells, f) == cells;
        int64_t expect = 0;
        for (size_t s = 0; ok && s < n; s++) {          /* offsets must be the running sum of the triangles */
            int o = tmp.order[s];
            ok = o >= 1 && o <= SAT_MAXDIM && tmp.cell_off[s] == expect && tmp.name[s * (SAT_LABELSIZE + 1) + SAT_LABELSIZE] == '\0';
            expect += (int64_t)o * (o + 1) / 2;
        }
        ok = ok && expect == hdr[1];
        tmp.count = tmp.capacity = (int)n;
        tmp.cells = tmp.cells_cap = (int64_t)cells;
    }
    fclose(f);
    if (!ok) {
        sat_set_free(&tmp);
        return -1;
    }
    sat_set_free(set);
    *set = tmp;
    return 0;
}
