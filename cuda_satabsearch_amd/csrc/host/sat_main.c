/*
 * sat_main.c - the `satabsearch` command line: drop-in for `cudaSaTabsearch`.
 *
 * Same options, stdin grammar and stdout bytes as the reference's main
 * (nvcc_src_current/cudaSaTabsearch.cu): options -c -q -r :605-626; "-q" SID list
 * :631-664 (SIDs cut to 7 chars, options fixed T T F); inline mode header :667-694;
 * LTYPE forced to T :696-700; SID lookup small class first :746-780; output = all
 * queries over the small class (order <= 96), then all queries over the large class
 * (97..111), three '#' header lines per (query, class) :1027-1030, rows :1102-1114 and
 * :1255-1268 (the large pass of the GPU path prints two blanks before the p-value).
 *
 * Host code is plain C; the search itself goes through the C ABI of
 * include/satabsearch.h (HIP kernel).  The reference is single-GPU (its TODO, :790);
 * here the database is sharded contiguously, by cost, over the visible GPUs (-g N) behind
 * the sat_multi_* entry points: launch on all, one RCCL gather to device 0, one copy to
 * the host, rows in file order.  Results do not depend on N (the random streams are keyed
 * by db ordinal).
 *
 * -c selects the host mode (csrc/host/sat_host_search.c): one CPU thread, one
 * sequential drand48 stream, byte-identical to the reference's -c.  It is never a
 * fallback: without -c a missing GPU is an error.
 *
 * Extensions: -g N (GPUs to use; default 1 as the reference, 0 = all visible), -G 0,2,3 (which GPUs; a GPU named twice
 * holds two shards), -s SEED (Philox seed, default 1234),
 * -k K (print only the K best rows per query, ranked on the GPU by raw score, ties in
 * database order - the `sort -k 2,2nr | head` users run on the reference's output),
 * -b (keep a binary image `dbfile.satbin` beside the database and load it instead of
 * parsing when it is newer than the ASCII file).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "satabsearch.h"
#include "sat_gumbel.h"
#include "sat_host_search.h"
#include "sat_parse.h"


static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static void usage(const char *prog)
{
    fprintf(stderr, "Usage: %s [-c] [-q dbfile] [-r restarts] [-g gpus] [-G gpu,gpu,...] [-s seed] [-k K] [-b]\n", prog);
    fprintf(stderr, "  -c : run on host CPU not GPU card\n");
    fprintf(stderr, "  -q dbfile : database is read from dbfile, list of query\n"
                    "              ids is read from stdin\n");
    fprintf(stderr, "  -r restarts : number of restarts. Default %d\n", 128);
    fprintf(stderr, "  -g gpus : number of GPUs to shard the database over (0 = all visible). Default 1\n");
    fprintf(stderr, "  -G list : the GPUs to use, e.g. 0,2,3 (a GPU named twice holds two shards)\n");
    fprintf(stderr, "  -s seed : seed of the GPU random streams. Default %d\n", SAT_DEFAULT_SEED);
    fprintf(stderr, "  -k K : print only the K best rows per query (GPU mode)\n");
    fprintf(stderr, "  -b : cache the parsed database as dbfile.satbin\n");
    exit(1);
}

/* ---- stdout.  A query list against a database is millions of rows "name score norm2 z p": formatted with
 * printf they take several times the GPU search (3 M rows: 1.5 s against 0.25 s).  The rows are assembled
 * in a buffer instead, from pieces that printf itself produced: the "%g %g" text of (z, p) is cached per
 * truncated norm2 score (z and p are functions of that integer, sat_gumbel.h), the "%g" text of norm2 per
 * (score, n1 + n2) - byte-identical to printf("%-8s %d %g %g %g\n") by construction. */
static char out_buf[1 << 20];
static size_t out_len = 0;

static void out_flush(void)
{
    if (out_len) fwrite(out_buf, 1, out_len, stdout);
    out_len = 0;
    fflush(stdout);
}

static inline void out_bytes(const char *p, size_t n)
{
    if (out_len + n > sizeof(out_buf)) out_flush();
    if (n > sizeof(out_buf)) { fwrite(p, 1, n, stdout); return; }
    memcpy(out_buf + out_len, p, n);
    out_len += n;
}

static void print_header(int ltype, int lorder, int lsoln, const char *qid, const char *dbfile)
{
    char line[SAT_MAX_LINE_LEN + 64];
    int n = snprintf(line, sizeof line, "# cudaSaTabsearch LTYPE = %c LORDER = %c LSOLN = %c\n",
                     ltype ? 'T' : 'F', lorder ? 'T' : 'F', lsoln ? 'T' : 'F');
    out_bytes(line, (size_t)n);
    n = snprintf(line, sizeof line, "# QUERY ID = %-8s\n", qid);
    out_bytes(line, (size_t)n);
    n = snprintf(line, sizeof line, "# DBFILE = %-80s\n", dbfile);
    out_bytes(line, (size_t)n);
}

/* cached printf("%g") texts */
#define ZP_SLOTS 512                       /* truncated norm2 score -256 .. 255 */
#define N2_SCORE_LO (-256)
#define N2_SCORE_N 1280                    /* scores -256 .. 1023 */
#define N2_SUM_N (2 * SAT_MAXDIM + 1)      /* n1 + n2 */
/* (a text that does not fit its slot is never cached - len stays 0 - and is formatted again row by row) */
typedef struct { unsigned char len; char text[23]; } g_text;          /* "%g" of a double: at most 13 characters */
typedef struct { unsigned char len; char text[47]; } zp_text;         /* " %g  %g\n": at most 30 */
static zp_text zp_cache[2][ZP_SLOTS];      /* [wide gap]["z p" of the integer] */
static g_text *norm2_cache = NULL;         /* [score - lo][n1 + n2], allocated on first use */

static inline void out_int(int v)
{
    char tmp[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) tmp[n++] = '-';
    if (out_len + 12 > sizeof(out_buf)) out_flush();
    while (n) out_buf[out_len++] = tmp[--n];
}

static void out_row(const char *name, int score, double norm2score, double zscore, double pvalue, int sum, int wide_gap,
                    int stats_from_host)
{
    /* "%-8s " */
    char nm[9];
    size_t ln = strlen(name);
    if (ln > 8) {                                          /* not produced by the reader; printf prints it whole */
        out_bytes(name, ln);
    } else {
        memset(nm, ' ', 8);
        memcpy(nm, name, ln);
        out_bytes(nm, 8);
    }
    out_bytes(" ", 1);
    out_int(score);
    out_bytes(" ", 1);
    /* norm2 */
    const int si = score - N2_SCORE_LO;
    if (si >= 0 && si < N2_SCORE_N && sum >= 0 && sum < N2_SUM_N) {
        if (!norm2_cache) norm2_cache = (g_text *)calloc((size_t)N2_SCORE_N * N2_SUM_N, sizeof(g_text));
        g_text *c = norm2_cache ? &norm2_cache[(size_t)si * N2_SUM_N + sum] : NULL;
        if (c && !c->len) {
            const int n = snprintf(c->text, sizeof c->text, "%g", norm2score);
            if (n > 0 && (size_t)n < sizeof c->text) c->len = (unsigned char)n;
        }
        if (c && c->len) out_bytes(c->text, c->len);
        else { char t[32]; out_bytes(t, (size_t)snprintf(t, sizeof t, "%g", norm2score)); }
    } else {
        char t[32];
        out_bytes(t, (size_t)snprintf(t, sizeof t, "%g", norm2score));
    }
    /* " z p\n": a function of the truncated norm2 score */
    const int x = (int)norm2score;
    if (stats_from_host && x >= -256 && x < 256) {
        zp_text *c = &zp_cache[wide_gap ? 1 : 0][x + 256];
        if (!c->len) {
            const int n = snprintf(c->text, sizeof c->text, wide_gap ? " %g  %g\n" : " %g %g\n", zscore, pvalue);
            if (n > 0 && (size_t)n < sizeof c->text) c->len = (unsigned char)n;
        }
        if (c->len) out_bytes(c->text, c->len);
        else { char t[64]; out_bytes(t, (size_t)snprintf(t, sizeof t, wide_gap ? " %g  %g\n" : " %g %g\n", zscore, pvalue)); }
    } else {
        char t[64];
        out_bytes(t, (size_t)snprintf(t, sizeof t, wide_gap ? " %g  %g\n" : " %g %g\n", zscore, pvalue));
    }
}

static inline void out_map_line(int a, int b)
{
    /* "%3d %3d\n" for 1 <= a, b <= 999 */
    char t[8];
    t[0] = a >= 100 ? (char)('0' + a / 100) : ' ';
    t[1] = a >= 10 ? (char)('0' + a / 10 % 10) : ' ';
    t[2] = (char)('0' + a % 10);
    t[3] = ' ';
    t[4] = b >= 100 ? (char)('0' + b / 100) : ' ';
    t[5] = b >= 10 ? (char)('0' + b / 10 % 10) : ' ';
    t[6] = (char)('0' + b % 10);
    t[7] = '\n';
    out_bytes(t, 8);
}

static void print_row(const char *name, int score, int n1, int n2, const int32_t *map, int lsoln,
                      int wide_gap)
{
    double norm2score = sat_norm2(score, n1, n2);
    /* z and p only when their cached text is missing */
    const int x = (int)norm2score;
    double zscore = 0.0, pvalue = 0.0;
    if (!(x >= -256 && x < 256 && zp_cache[wide_gap ? 1 : 0][x + 256].len)) {
        zscore = sat_z_gumbel_trunc(norm2score);
        pvalue = sat_pv_gumbel(zscore);
    }
    out_row(name, score, norm2score, zscore, pvalue, n1 + n2, wide_gap, 1);
    if (lsoln)
        for (int k = 0; k < n1; k++)
            if (map[k] >= 0)
                out_map_line(k + 1, map[k] + 1);
}

int main(int argc, char *argv[])
{
    char dbfile[SAT_MAX_LINE_LEN] = "";
    char buf[SAT_MAX_LINE_LEN];
    int use_gpu = 1, querydbmode = 0, maxstart = 128, want_gpus = 1, bincache = 0, topk = 0;
    unsigned long long seed = SAT_DEFAULT_SEED;
    int ltype = 0, lorder = 0, lsoln = 0;
    char cltype = 'F', clorder = 'F', clsoln = 'F';
    int c;

    int dev_list[64], ndev_list = 0;
    while ((c = getopt(argc, argv, "cq:r:g:G:s:bk:")) != -1) {
        switch (c) {
        case 'c': use_gpu = 0; break;
        case 'q': querydbmode = 1; strncpy(dbfile, optarg, sizeof(dbfile) - 1); break;
        case 'r': maxstart = atoi(optarg); break;
        case 'g': want_gpus = atoi(optarg); break;
        case 'G':
            for (char *tok = strtok(optarg, ","); tok && ndev_list < 64; tok = strtok(NULL, ","))
                dev_list[ndev_list++] = atoi(tok);
            break;
        case 's': seed = strtoull(optarg, NULL, 0); break;
        case 'b': bincache = 1; break;
        case 'k': topk = atoi(optarg); break;
        default: usage(argv[0]);
        }
    }
    fprintf(stderr, "MAXDIM = %d\n", SAT_MAXDIM);
    atexit(out_flush);                                   /* every exit path, exit(1) included */

    sat_struct_set queries, db;
    sat_set_init(&queries);
    sat_set_init(&db);
    char *sid_list = NULL;
    int num_queries = 0;

    if (querydbmode) {
        cltype = 'T'; ltype = 1;
        clorder = 'T'; lorder = 1;
        clsoln = 'F'; lsoln = 0;
        while (!feof(stdin)) {
            if (!fgets(buf, SAT_MAX_LINE_LEN, stdin))
                break;
            char *grown = (char *)realloc(sid_list, (size_t)(num_queries + 1) * (SAT_LABELSIZE + 1));
            if (!grown) { fprintf(stderr, "realloc queryid_list failed\n"); exit(1); }
            sid_list = grown;
            char *sid = sid_list + (size_t)num_queries * (SAT_LABELSIZE + 1);
            memset(sid, 0, SAT_LABELSIZE + 1);
            strncpy(sid, buf, SAT_LABELSIZE);
            sid[SAT_LABELSIZE - 1] = '\0';
            size_t len = strlen(sid);
            if (len && sid[len - 1] == '\n') sid[len - 1] = '\0';
            num_queries++;
        }
    } else {
        if (fscanf(stdin, "%s\n", dbfile) != 1) {
            fprintf(stderr, "ERROR reading dbfilename from stdin\n");
            exit(1);
        }
        if (fscanf(stdin, "%c %c %c\n", &cltype, &clorder, &clsoln) != 3) {
            fprintf(stderr, "ERROR reading options from stdin\n");
            exit(1);
        }
        ltype = cltype == 'T';
        lorder = clorder == 'T';
        lsoln = clsoln == 'T';
        num_queries = sat_read_structures(stdin, &queries, "query");
        if (num_queries < 0) {
            fprintf(stderr, "ERROR loading query structures from stdin\n");
            exit(1);
        } else if (num_queries == 0) {
            fprintf(stderr, "ERROR: no query structures found on stdin\n");
            exit(1);
        }
        fprintf(stderr, "Read %d query structures\n", num_queries);
    }
    if (!ltype) {
        fprintf(stderr, "WARNING: LTYPE is always set to T\n");
        ltype = 1;
    }

    FILE *dbfp = fopen(dbfile, "r");
    if (!dbfp) {
        fprintf(stderr, "ERROR opening db file %s\n", dbfile);
        exit(1);
    }
    fclose(dbfp);
    fprintf(stderr, "Loading database...\n");
    double t0 = now_ms();
    int total = -1;
    char binpath[SAT_MAX_LINE_LEN + 16];
    snprintf(binpath, sizeof(binpath), "%s.satbin", dbfile);
    if (bincache) {
        struct stat sa, sb;
        if (stat(dbfile, &sa) == 0 && stat(binpath, &sb) == 0 && sb.st_mtime >= sa.st_mtime &&
            sat_set_load_binary(binpath, &db) == 0) {
            total = db.count;
            fprintf(stderr, "(binary image %s)\n", binpath);
        }
    }
    if (total < 0) {
        total = sat_read_structures_file(dbfile, &db, "database");     /* mmap reader, same semantics */
        if (total >= 0 && bincache && sat_set_save_binary(&db, binpath) != 0)
            fprintf(stderr, "WARNING: could not write %s\n", binpath);
    }
    if (total < 0) {
        fprintf(stderr, "ERROR loading database\n");
        exit(1);
    }
    /* the two passes of the reference: small class then large class, file order inside */
    int *cls_index[2], cls_count[2] = { 0, 0 };
    cls_index[0] = (int *)malloc(sizeof(int) * (size_t)(total + 1));
    cls_index[1] = (int *)malloc(sizeof(int) * (size_t)(total + 1));
    if (!cls_index[0] || !cls_index[1]) { fprintf(stderr, "malloc failed\n"); exit(1); }
    for (int s = 0; s < db.count; s++) {
        int k = db.order[s] > SAT_MAXDIM_SMALL;
        cls_index[k][cls_count[k]++] = s;
    }
    fprintf(stderr, "Loaded %d db entries (%d order > %d) in %f ms\n",
            total, cls_count[1], SAT_MAXDIM_SMALL, now_ms() - t0);
    if (total == 0) {
        fprintf(stderr, "ERROR: empty database\n");
        exit(1);
    }

    /* -q: SID -> db structure (small class searched first) */
    const sat_struct_set *qsrc = querydbmode ? &db : &queries;
    int *qindex = (int *)malloc(sizeof(int) * (size_t)(num_queries + 1));
    for (int i = 0; i < num_queries; i++) {
        qindex[i] = i;
        if (!querydbmode)
            continue;
        const char *sid = sid_list + (size_t)i * (SAT_LABELSIZE + 1);
        int found = -1;
        for (int k = 0; k < 2 && found < 0; k++)
            for (int d = 0; d < cls_count[k]; d++)
                if (!strcasecmp(sid, sat_set_name(&db, cls_index[k][d]))) {
                    found = cls_index[k][d];
                    break;
                }
        if (found < 0) {
            fprintf(stderr, "ERROR: query %s not found\n", sid);
            exit(1);
        }
        qindex[i] = found;
    }
    fprintf(stderr, "maxstart = %d\n", maxstart);

    int32_t *scores = (int32_t *)malloc(sizeof(int32_t) * (size_t)total);
    int32_t *ssemaps = lsoln ? (int32_t *)malloc(sizeof(int32_t) * SAT_MAXDIM * (size_t)total) : NULL;
    if (!scores || (lsoln && !ssemaps)) { fprintf(stderr, "malloc scores failed\n"); exit(1); }

    if (!use_gpu) {
        /* ---- host mode: class by class, query by query, ONE stream for everything ---- */
        sat_host_stream stream;
        sat_host_stream_seed(&stream, 1234);
        for (int k = 0; k < 2; k++) {
            if (k == 1 && cls_count[1] == 0)
                break;
            for (int qi = 0; qi < num_queries; qi++) {
                const int qs = qindex[qi], n1 = qsrc->order[qs];
                print_header(ltype, lorder, lsoln, sat_set_name(qsrc, qs), dbfile);
                fprintf(stderr, "Executing simulated annealing tableaux match kernel on host for query %s...\n",
                        sat_set_name(qsrc, qs));
                double t1 = now_ms();
                if (sat_host_search(&db, cls_index[k], cls_count[k], qsrc, qs, lorder, lsoln, maxstart,
                                    &stream, scores, ssemaps) != 0) {
                    fprintf(stderr, "malloc failed in host search\n");
                    exit(1);
                }
                double ms = now_ms() - t1;
                fprintf(stderr, "host execution time %f ms\n", ms);
                fprintf(stderr, "%f million iterations/sec\n",
                        ((double)cls_count[k] * ((double)maxstart * SAT_MAXITER) / (ms / 1000)) / 1.0e6);
                for (int d = 0; d < cls_count[k]; d++)
                    print_row(sat_set_name(&db, cls_index[k][d]), scores[d], n1, db.order[cls_index[k][d]],
                              ssemaps ? ssemaps + (size_t)d * SAT_MAXDIM : NULL, lsoln, 0);
            }
        }
        /* main owns every host array (as H.cu:1314-1324) */
        free(scores);
        free(ssemaps);
        free(qindex);
        free(sid_list);
        free(cls_index[0]);
        free(cls_index[1]);
        free(norm2_cache);
        sat_set_free(&queries);
        sat_set_free(&db);
        return 0;
    }

    /* ---- GPU mode ---- */
    int ndev = sat_device_count();
    if (ndev <= 0) {
        fprintf(stderr, "There is no usable HIP device (use -c for the host mode).\n");
        exit(1);
    }
    fprintf(stderr, "found %d HIP devices\n", ndev);
    int ngpu = want_gpus > 0 ? want_gpus : ndev;
    if (ngpu > ndev) ngpu = ndev;
    if (ndev_list > 0) ngpu = ndev_list;
    if (ngpu > total) ngpu = total;

    /* One multi-GPU context: the database is cut into contiguous shards of equal COST (entries of a
     * size-sorted database differ several-fold in cost, sat_shard.h), every GPU holds its shard, a
     * search is queued on all of them and one gather (RCCL over xGMI) brings the rows to device 0. */
    t0 = now_ms();
    sat_multi *multi = sat_multi_create(ngpu, ndev_list > 0 ? dev_list : NULL, seed);
    if (!multi) {
        fprintf(stderr, "sat_multi_create(%d) failed: %s\n", ngpu, sat_last_error());
        exit(1);
    }
    if (sat_multi_db_upload_packed(multi, total, db.order, db.cell_off, db.tab, db.dist) != SAT_OK) {
        fprintf(stderr, "database upload failed: %s\n", sat_last_error());
        exit(1);
    }
    fprintf(stderr, "Copied %d entries to %d GPU(s) in %f ms (gather: %s)\n", total, ngpu, now_ms() - t0,
            sat_multi_gather_kind(multi));
    if (ngpu > 1) {
        int32_t *begin = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ngpu + 1));
        if (begin && sat_multi_shards(multi, begin) == SAT_OK)
            for (int g = 0; g < ngpu; g++)
                fprintf(stderr, "  GPU %d: entries %d .. %d\n", g, begin[g], begin[g + 1] - 1);
        free(begin);
    }

    /* rows of the large class are printed after every query's small-class block */
    int32_t *large_scores = NULL, *large_maps = NULL;
    if (cls_count[1] > 0 && topk <= 0) {
        large_scores = (int32_t *)malloc(sizeof(int32_t) * (size_t)cls_count[1] * num_queries);
        if (lsoln) large_maps = (int32_t *)malloc(sizeof(int32_t) * SAT_MAXDIM * (size_t)cls_count[1] * num_queries);
        if (!large_scores || (lsoln && !large_maps)) { fprintf(stderr, "malloc failed\n"); exit(1); }
    }

    /* Queries go to the GPUs in batches: one set of launches scores a whole batch (grid =
     * entries x queries), which is what fills the machine when the database is small and the
     * query list long (-q).  The batch size is bounded by the host result buffers. */
    int batch = 256;
    if (topk <= 0) {
        const size_t per_query = (size_t)total * (lsoln ? (SAT_MAXDIM + 1) : 1) * sizeof(int32_t);
        const size_t budget = (size_t)1 << 30;
        if ((size_t)batch * per_query > budget) batch = (int)(budget / per_query);
    }
    if (batch < 1) batch = 1;
    if (batch > num_queries) batch = num_queries;
    free(scores);
    free(ssemaps);
    scores = NULL;
    ssemaps = NULL;
    const int kk = topk < total ? topk : total;
    sat_hit *hits = NULL;
    int32_t *hit_maps = NULL;
    if (topk > 0) {
        /* best K per query: only K rows per query (and GPU) ever leave the GPUs */
        hits = (sat_hit *)malloc(sizeof(sat_hit) * (size_t)kk * batch);
        hit_maps = lsoln ? (int32_t *)malloc(sizeof(int32_t) * SAT_MAXDIM * (size_t)kk * batch) : NULL;
        if (!hits || (lsoln && !hit_maps)) { fprintf(stderr, "malloc failed\n"); exit(1); }
    } else {
        scores = (int32_t *)malloc(sizeof(int32_t) * (size_t)total * batch);
        ssemaps = lsoln ? (int32_t *)malloc(sizeof(int32_t) * SAT_MAXDIM * (size_t)total * batch) : NULL;
        if (!scores || (lsoln && !ssemaps)) { fprintf(stderr, "malloc failed\n"); exit(1); }
    }
    uint8_t *qtabs = (uint8_t *)calloc((size_t)batch * SAT_MAXDIM * SAT_MAXDIM, 1);
    float *qdmats = (float *)calloc((size_t)batch * SAT_MAXDIM * SAT_MAXDIM, sizeof(float));
    uint8_t *qtypes = (uint8_t *)calloc((size_t)batch * SAT_MAXDIM, 1);
    int32_t *n1s = (int32_t *)malloc(sizeof(int32_t) * (size_t)batch);
    if (!qtabs || !qdmats || !qtypes || !n1s) {
        fprintf(stderr, "malloc failed\n");
        exit(1);
    }
    int exit_status = 0;

    for (int q0 = 0; q0 < num_queries; q0 += batch) {
        const int nqb = num_queries - q0 < batch ? num_queries - q0 : batch;
        for (int b = 0; b < nqb; b++) {
            const int qs = qindex[q0 + b];
            n1s[b] = qsrc->order[qs];
            sat_set_expand(qsrc, qs, SAT_MAXDIM, qtabs + (size_t)b * SAT_MAXDIM * SAT_MAXDIM,
                           qdmats + (size_t)b * SAT_MAXDIM * SAT_MAXDIM);
            for (int i = 0; i < n1s[b]; i++)
                qtypes[(size_t)b * SAT_MAXDIM + i] = qtabs[(size_t)b * SAT_MAXDIM * SAT_MAXDIM + i * SAT_MAXDIM + i];
        }
        fprintf(stderr, "Executing simulated annealing tableaux match kernel on GPU for %d quer%s (from %s)...\n",
                nqb, nqb == 1 ? "y" : "ies", sat_set_name(qsrc, qindex[q0]));
        double ms = 0.0;
        int rc = sat_multi_queries_set(multi, nqb, n1s, qtabs, qdmats, SAT_MAXDIM, qtypes, (uint32_t)q0);
        if (rc == SAT_OK)
            rc = topk > 0 ? sat_multi_search_topk(multi, lorder, lsoln, maxstart, kk, hits, hit_maps, &ms)
                          : sat_multi_search(multi, lorder, lsoln, maxstart, scores, ssemaps, &ms);
        if (rc < 0) {
            fprintf(stderr, "kernel launch failed: %s\n", sat_last_error());
            exit_status = 1;
            goto bye;
        }
        fprintf(stderr, "GPU execution time %f ms\n", ms);
        fprintf(stderr, "%f million iterations/sec\n",
                ((double)total * nqb * ((double)maxstart * SAT_MAXITER) / (ms / 1000)) / 1.0e6);
        if (topk > 0) {
            for (int b = 0; b < nqb; b++) {
                const int qs = qindex[q0 + b], n1 = n1s[b];
                print_header(ltype, lorder, lsoln, sat_set_name(qsrc, qs), dbfile);
                for (int r = 0; r < rc; r++) {
                    const sat_hit *h = hits + (size_t)b * rc + r;
                    out_row(sat_set_name(&db, h->entry), h->score, h->norm2, h->zscore, h->pvalue, n1 + db.order[h->entry], 0, 0);
                    if (lsoln) {
                        const int32_t *map = hit_maps + ((size_t)b * rc + r) * SAT_MAXDIM;
                        for (int k2 = 0; k2 < n1; k2++)
                            if (map[k2] >= 0)
                                out_map_line(k2 + 1, map[k2] + 1);
                    }
                }
            }
            continue;
        }
        for (int b = 0; b < nqb; b++) {
            const int qi = q0 + b, qs = qindex[qi], n1 = n1s[b];
            const int32_t *qscores = scores + (size_t)b * total;
            const int32_t *qmaps = ssemaps ? ssemaps + (size_t)b * total * SAT_MAXDIM : NULL;
            print_header(ltype, lorder, lsoln, sat_set_name(qsrc, qs), dbfile);
            for (int d = 0; d < cls_count[0]; d++) {
                int s = cls_index[0][d];
                print_row(sat_set_name(&db, s), qscores[s], n1, db.order[s],
                          qmaps ? qmaps + (size_t)s * SAT_MAXDIM : NULL, lsoln, 0);
            }
            for (int d = 0; d < cls_count[1]; d++) {
                int s = cls_index[1][d];
                large_scores[(size_t)qi * cls_count[1] + d] = qscores[s];
                if (lsoln)
                    memcpy(large_maps + ((size_t)qi * cls_count[1] + d) * SAT_MAXDIM,
                           qmaps + (size_t)s * SAT_MAXDIM, sizeof(int32_t) * SAT_MAXDIM);
            }
        }
    }
    if (cls_count[1] > 0 && topk <= 0)
        for (int qi = 0; qi < num_queries; qi++) {
            const int qs = qindex[qi], n1 = qsrc->order[qs];
            print_header(ltype, lorder, lsoln, sat_set_name(qsrc, qs), dbfile);
            for (int d = 0; d < cls_count[1]; d++) {
                int s = cls_index[1][d];
                print_row(sat_set_name(&db, s), large_scores[(size_t)qi * cls_count[1] + d], n1, db.order[s],
                          lsoln ? large_maps + ((size_t)qi * cls_count[1] + d) * SAT_MAXDIM : NULL, lsoln, 1);
            }
        }
bye:
    fprintf(stderr, "copied %llu bytes of results from the GPU(s)\n", sat_multi_stat_d2h_bytes(multi));
    sat_multi_destroy(multi);
    (void)cltype; (void)clorder; (void)clsoln;
    free(scores);
    free(ssemaps);
    free(hits);
    free(hit_maps);
    free(large_scores);
    free(large_maps);
    free(qtabs);
    free(qdmats);
    free(qtypes);
    free(n1s);
    free(qindex);
    free(sid_list);
    free(cls_index[0]);
    free(cls_index[1]);
    free(norm2_cache);
    sat_set_free(&queries);
    sat_set_free(&db);
    return exit_status;
}
