/*
 * sat_parse.h - ASCII tableau + SSE-distance-matrix reader (host side, plain C).
 *
 * Same input grammar and the same fixed-column cell semantics as the reference
 * reader (nvcc_src_current/parsetableaux.c:193-227 parse_tableau, :276-294
 * parse_distmatrix, :317-506 read_database, :522-632 read_queries), but a
 * different in-memory model: the reference keeps every structure in a dense
 * padded 96x96 or 111x111 slot (46 KB / 61 KB per entry); here every structure
 * is stored once as its packed lower triangle (1 B code + 4 B distance per
 * cell, diagonal included), which is also the layout uploaded to HBM.
 *
 * Lower-triangle cell (i,j), j <= i, of structure s lives at
 *     set->cell_off[s] + i*(i+1)/2 + j
 * in set->tab (code bytes) and set->dist (floats).
 */
#ifndef SAT_PARSE_H
#define SAT_PARSE_H

#include <stdio.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAT_MAXDIM       111   /* saparams.h:15  largest structure accepted          */
#define SAT_MAXDIM_SMALL  96   /* saparams.h:17  "small" class limit (MAXDIM_GPU)      */
#define SAT_LABELSIZE      8   /* saparams.h:24                                        */
#define SAT_MAX_LINE_LEN 2048  /* parsetableaux.h                                      */

/* encodings, parsetableaux.c:13-33 */
#define SAT_SSE_E  0
#define SAT_SSE_XA 1
#define SAT_SSE_XI 2
#define SAT_SSE_XG 3

typedef struct sat_struct_set {
    int       count;      /* structures held                                          */
    int       capacity;
    int64_t   cells;      /* total lower-triangle cells held                          */
    int64_t   cells_cap;
    int      *order;      /* [count]  number of SSEs                                  */
    char     *name;       /* [count][SAT_LABELSIZE+1], NUL terminated                 */
    int64_t  *cell_off;   /* [count]  first cell of structure s                       */
    uint8_t  *tab;        /* [cells]  packed code bytes (diagonal = SSE type)         */
    float    *dist;       /* [cells]  distances in Angstrom (diagonal = type as x.000)*/
    int       skipped;    /* structures dropped: order > SAT_MAXDIM or order < 1       */
} sat_struct_set;

void sat_set_init(sat_struct_set *set);
void sat_set_free(sat_struct_set *set);

/* Append one structure given as packed lower triangles; returns its index or -1. */
int sat_set_append(sat_struct_set *set, const char *name, int order,
                   const uint8_t *tab_tri, const float *dist_tri);

/*
 * Read every "name order / tableau rows / distance rows" record up to EOF.
 * `what` is "database" or "query" (only used in the warnings, which are worded
 * as parsetableaux.c:459, 500, 603, 627).  Returns the number of structures
 * appended, or -1 on allocation failure.  Unknown code letters terminate the
 * process with exit(1), as the reference does (parsetableaux.c:70-73,109-112).
 */
int sat_read_structures(FILE *fp, sat_struct_set *set, const char *what);

/*
 * Same grammar and cell semantics as sat_read_structures, from a memory image of the file
 * (mmap): no stdio, and a fast path for the "%6.3f" distance cells the database builder
 * writes; any cell that does not look like [ddd].ddd goes through strtof() as before.
 * ~10x faster ingest for million-entry databases.
 */
int sat_read_structures_mem(const char *text, size_t len, sat_struct_set *set, const char *what);

/*
 * The same on `nthreads` threads: the image is cut at record headers near the equal-size marks, the
 * pieces are parsed concurrently and concatenated in file order.  A piece that does not end exactly at
 * its cut (a record claiming more rows than it has) makes the whole image go through the sequential
 * reader again, whose result is the definition.  Per-record warnings may interleave on stderr.
 */
int sat_read_structures_mem_mt(const char *text, size_t len, sat_struct_set *set, const char *what, int nthreads);

/* One distance cell exactly as the memory-image reader parses it (exposed for the tests that
 * compare the fast path with strtof over its whole domain). */
float sat_distance_cell(const char *text);

/* mmap `path` and parse it with sat_read_structures_mem_mt on up to 16 threads (SAT_PARSE_THREADS
 * overrides the count; files under 2 MB are parsed sequentially); -1 if the file cannot be mapped. */
int sat_read_structures_file(const char *path, sat_struct_set *set, const char *what);

/*
 * Binary image of a structure set (the packed arrays as they are, little endian):
 * lets a command line skip the ASCII parse on later runs.  Returns 0 / -1.
 * sat_set_load_binary refuses files whose header, sizes or offsets are inconsistent.
 */
int sat_set_save_binary(const sat_struct_set *set, const char *path);
int sat_set_load_binary(const char *path, sat_struct_set *set);

/*
 * Write the set in the ASCII format the readers above parse (and the reference's database
 * builder scripts/convdb2.py:214-226 emits): "%-8s %4d" header, tableau rows, "%6.3f " distance
 * rows, a blank line after each record.  Returns 0 / -1.
 */
int sat_set_write_ascii(const sat_struct_set *set, const char *path);

/* Expand structure s to dense symmetric pitch x pitch arrays (row-major). */
void sat_set_expand(const sat_struct_set *set, int s, int pitch,
                    uint8_t *tab_dense, float *dist_dense);

static inline const char *sat_set_name(const sat_struct_set *set, int s)
{
    return set->name + (size_t)s * (SAT_LABELSIZE + 1);
}

#ifdef __cplusplus
}
#endif
#endif /* SAT_PARSE_H */
