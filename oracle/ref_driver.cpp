/*
 * ref_driver.cpp - a small main() around the REFERENCE's own host objects
 * (TEST INFRASTRUCTURE; own code, contains nothing copied from the reference).
 *
 * oracle/Makefile compiles the reference's cudaSaTabsearch_kernel.cu (host
 * build, no -DCUDA), parsetableaux.c and gumbelstats.c straight from
 * /root/reference into oracle/_ref/ and links them with this file.  The
 * reference's own main (cudaSaTabsearch.cu) cannot be compiled here: it needs
 * the CUDA SDK sample headers (helper_cuda.h, helper_timer.h), curand and the
 * <<<>>> launch syntax.  This driver replays only its "-c" branch:
 *   stdin header        cudaSaTabsearch.cu:667-682
 *   read_queries        :684      (REF_CUDA5: one query, :parse_tableau/parse_distmatrix)
 *   LTYPE forced        :696-700
 *   read_database       :712-716
 *   srand48(1234)       :871
 *   small then large    :1276-1309
 *   per-query body      :308-459 (tabsearch_host_thread)
 *   -q dbfile           :631-664 (SIDs on stdin, cut to 7 characters, options fixed T T F),
 *                       :746-780 (case-insensitive lookup, small class first),
 *                       :356-400 (the query is copied out of the database arrays; a small-class
 *                       member is re-pitched 96 -> 111 cell by cell, off-diagonal cells only)
 * so that oracle/_ref/ref_oracle < X.input is what "cudaSaTabsearch -c < X.input"
 * prints on stdout, and ref_oracle -q db < sids what "cudaSaTabsearch -c -q db < sids" does.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>
#include <ctime>
#include <unistd.h>
#include <driver_types.h>

#include "saparams.h"
#include "parsetableaux.h"
#include "gumbelstats.h"

/* host symbols of the reference kernel translation unit (C++ linkage) */
extern int   c_qn_host;
extern char  c_qtab_host[MAXDIM * MAXDIM];
extern float c_qdmat_host[MAXDIM * MAXDIM];
extern char  c_qssetypes_host[MAXDIM];
void sa_tabsearch_host(int dbsize, int lorder, int lsoln, int maxstart,
                       cudaPitchedPtr d_tableaux, cudaExtent tableaux_extent,
                       int *d_orders,
                       cudaPitchedPtr d_distmatrices, cudaExtent distmatrices_extent,
                       int *outscore, int *outssemap, int *state);

static double now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static char dbfile[MAX_LINE_LEN];

/* -q mode: where query qi lives in the database arrays */
struct QueryRef { int large, index; };
static QueryRef *g_qref = NULL;
static char *g_small_tabs, *g_large_tabs, *g_small_names, *g_large_names;
static float *g_small_dmats, *g_large_dmats;
static int *g_small_orders, *g_large_orders;

static void run_class(int maxdim, int dbsize, char *tabs, float *dmats, int *orders, char *names,
                      int nq, char *qtabs, float *qdmats, int *qorders, char *qnames,
                      int ltype, int lorder, int lsoln, int maxstart)
{
    int *scores = (int *)malloc(sizeof(int) * (dbsize + 1));
    int *ssemaps = (int *)malloc(sizeof(int) * MAXDIM * (size_t)(dbsize + 1));
    cudaExtent text, dext;
    cudaPitchedPtr tpp, dpp;
    text.width = maxdim; text.height = maxdim; text.depth = dbsize;
    tpp.ptr = tabs; tpp.pitch = maxdim; tpp.xsize = maxdim; tpp.ysize = dbsize;
    dext.width = maxdim * sizeof(float); dext.height = maxdim; dext.depth = maxdim;
    dpp.ptr = dmats; dpp.pitch = maxdim * sizeof(float); dpp.xsize = maxdim; dpp.ysize = maxdim;

    for (int qi = 0; qi < nq; qi++) {
        char qid[LABELSIZE + 1];
        memset(qid, 0, sizeof(qid));
        if (g_qref && g_qref[qi].large) {
            const int d = g_qref[qi].index;
            strncpy(qid, g_large_names + d * (LABELSIZE + 1), LABELSIZE);
            c_qn_host = g_large_orders[d];
            memcpy(c_qtab_host, g_large_tabs + (size_t)d * MAXDIM * MAXDIM, sizeof(c_qtab_host));
            memcpy(c_qdmat_host, g_large_dmats + (size_t)d * MAXDIM * MAXDIM, sizeof(c_qdmat_host));
            for (int i = 0; i < c_qn_host; i++)
                c_qssetypes_host[i] = (g_large_tabs + (size_t)d * MAXDIM * MAXDIM)[i * MAXDIM + i];
        } else if (g_qref) {
            /* small-class member: off-diagonal cells only, pitch MAXDIM_GPU -> MAXDIM; whatever else
             * the query globals held stays */
            const int d = g_qref[qi].index;
            strncpy(qid, g_small_names + d * (LABELSIZE + 1), LABELSIZE);
            c_qn_host = g_small_orders[d];
            const char *t = g_small_tabs + (size_t)d * MAXDIM_GPU * MAXDIM_GPU;
            const float *dm = g_small_dmats + (size_t)d * MAXDIM_GPU * MAXDIM_GPU;
            for (int i = 0; i < c_qn_host; i++)
                for (int j = i + 1; j < c_qn_host; j++) {
                    c_qtab_host[i * MAXDIM + j] = c_qtab_host[j * MAXDIM + i] = t[i * MAXDIM_GPU + j];
                    c_qdmat_host[i * MAXDIM + j] = c_qdmat_host[j * MAXDIM + i] = dm[i * MAXDIM_GPU + j];
                }
            for (int i = 0; i < c_qn_host; i++)
                c_qssetypes_host[i] = t[i * MAXDIM_GPU + i];
        } else {
            strncpy(qid, qnames + qi * (LABELSIZE + 1), LABELSIZE);
            c_qn_host = qorders[qi];
            memcpy(c_qtab_host, qtabs + (size_t)qi * MAXDIM * MAXDIM, sizeof(c_qtab_host));
            memcpy(c_qdmat_host, qdmats + (size_t)qi * MAXDIM * MAXDIM, sizeof(c_qdmat_host));
            for (int i = 0; i < qorders[qi]; i++)
                c_qssetypes_host[i] = (qtabs + (size_t)qi * MAXDIM * MAXDIM)[i * MAXDIM + i];
        }

        printf("# cudaSaTabsearch LTYPE = %c LORDER = %c LSOLN = %c\n",
               ltype ? 'T' : 'F', lorder ? 'T' : 'F', lsoln ? 'T' : 'F');
        printf("# QUERY ID = %-8s\n", qid);
        printf("# DBFILE = %-80s\n", dbfile);

        int state = 0;
        double t0 = now_ms();
        sa_tabsearch_host(dbsize, lorder, lsoln, maxstart, tpp, text, orders, dpp, dext,
                          scores, ssemaps, &state);
        double ms = now_ms() - t0;
        fprintf(stderr, "host execution time %f ms\n", ms);
        fprintf(stderr, "%f million iterations/sec\n",
                ((double)dbsize * ((double)maxstart * MAXITER) / (ms / 1000)) / 1.0e6);

        for (int i = 0; i < dbsize; i++) {
            double norm2score = norm2(scores[i], c_qn_host, orders[i]);
            double zscore = z_gumbel(norm2score, gumbel_a, gumbel_b);
            double pvalue = pv_gumbel(zscore);
            printf("%-8s %d %g %g %g\n", names + i * (LABELSIZE + 1), scores[i],
                   norm2score, zscore, pvalue);
            if (lsoln)
                for (int k = 0; k < c_qn_host; k++)
                    if (ssemaps[i * MAXDIM + k] >= 0)
                        printf("%3d %3d\n", k + 1, ssemaps[i * MAXDIM + k] + 1);
        }
    }
    free(scores);
    free(ssemaps);
}

int main(int argc, char *argv[])
{
    int maxstart = DEFAULT_MAXSTART;
    int c, querydbmode = 0;
    while ((c = getopt(argc, argv, "cr:q:")) != -1) {
        if (c == 'r') maxstart = atoi(optarg);
        else if (c == 'q') { querydbmode = 1; strncpy(dbfile, optarg, sizeof(dbfile) - 1); }
        else if (c != 'c') { fprintf(stderr, "usage: %s [-c] [-r restarts] [-q dbfile] < input\n", argv[0]); return 1; }
    }
    char cltype = 'T', clorder = 'T', clsoln = 'F';
    char *sids = NULL;
    int nsids = 0;
    if (querydbmode) {
        /* one SID per line, kept to LABELSIZE - 1 = 7 characters, trailing newline dropped */
        char line[MAX_LINE_LEN];
        while (!feof(stdin)) {
            if (!fgets(line, MAX_LINE_LEN, stdin)) break;
            sids = (char *)realloc(sids, (size_t)(nsids + 1) * (LABELSIZE + 1));
            char *sid = sids + (size_t)nsids * (LABELSIZE + 1);
            memset(sid, 0, LABELSIZE + 1);
            strncpy(sid, line, LABELSIZE);
            sid[LABELSIZE - 1] = '\0';
            if (strlen(sid) > 0 && sid[strlen(sid) - 1] == '\n') sid[strlen(sid) - 1] = '\0';
            nsids++;
        }
    } else {
        if (fscanf(stdin, "%s\n", dbfile) != 1) { fprintf(stderr, "ERROR reading dbfilename from stdin\n"); return 1; }
        if (fscanf(stdin, "%c %c %c\n", &cltype, &clorder, &clsoln) != 3) { fprintf(stderr, "ERROR reading options from stdin\n"); return 1; }
    }
    int ltype = cltype == 'T', lorder = clorder == 'T', lsoln = clsoln == 'T';

    char *qtabs = NULL; float *qdmats = NULL; int *qorders = NULL; char *qnames = NULL;
    int nq;
    if (querydbmode) nq = nsids;
    else {
#ifdef REF_CUDA5
    /* 2013 sources: exactly one query on stdin, header "%8s %d" then the two matrices */
    qtabs = (char *)calloc(MAXDIM * MAXDIM, 1);
    qdmats = (float *)calloc(MAXDIM * MAXDIM, sizeof(float));
    qorders = (int *)calloc(1, sizeof(int));
    qnames = (char *)calloc(LABELSIZE + 1, 1);
    if (fscanf(stdin, "%8s %d\n", qnames, &qorders[0]) != 2) { fprintf(stderr, "ERROR reading query header\n"); return 1; }
    parse_tableau(stdin, MAXDIM, qorders[0], qtabs);
    parse_distmatrix(stdin, MAXDIM, qorders[0], qdmats, 0);
    nq = 1;
#else
    nq = read_queries(stdin, &qtabs, &qdmats, &qorders, &qnames);
    if (nq <= 0) { fprintf(stderr, "ERROR: no query structures found on stdin\n"); return 1; }
#endif
    }
    if (!ltype) { fprintf(stderr, "WARNING: LTYPE is always set to T\n"); ltype = 1; }

    FILE *dbfp = fopen(dbfile, "r");
    if (!dbfp) { fprintf(stderr, "ERROR opening db file %s\n", dbfile); return 1; }
    char *tabs, *ltabs, *names, *lnames; float *dmats, *ldmats; int *orders, *lorders; int nlarge;
    int total = read_database(dbfp, &tabs, &dmats, &ltabs, &ldmats, &orders, &names, &lorders, &lnames, &nlarge);
    fclose(dbfp);
    if (total < 0) { fprintf(stderr, "ERROR loading database\n"); return 1; }
    fprintf(stderr, "Loaded %d db entries (%d order > %d)\n", total, nlarge, MAXDIM_GPU);
    fprintf(stderr, "maxstart = %d\n", maxstart);
    if (querydbmode) {
        g_small_tabs = tabs; g_small_dmats = dmats; g_small_orders = orders; g_small_names = names;
        g_large_tabs = ltabs; g_large_dmats = ldmats; g_large_orders = lorders; g_large_names = lnames;
        g_qref = (QueryRef *)malloc(sizeof(QueryRef) * (size_t)(nq + 1));
        for (int i = 0; i < nq; i++) {
            const char *sid = sids + (size_t)i * (LABELSIZE + 1);
            int found = 0;
            for (int j = 0; j < total - nlarge && !found; j++)
                if (!strcasecmp(sid, names + j * (LABELSIZE + 1))) { g_qref[i].large = 0; g_qref[i].index = j; found = 1; }
            for (int j = 0; j < nlarge && !found; j++)
                if (!strcasecmp(sid, lnames + j * (LABELSIZE + 1))) { g_qref[i].large = 1; g_qref[i].index = j; found = 1; }
            if (!found) { fprintf(stderr, "ERROR: query %s not found\n", sid); return 1; }
        }
    }

    srand48(1234);
    run_class(MAXDIM_GPU, total - nlarge, tabs, dmats, orders, names,
              nq, qtabs, qdmats, qorders, qnames, ltype, lorder, lsoln, maxstart);
    if (nlarge > 0)
        run_class(MAXDIM, nlarge, ltabs, ldmats, lorders, lnames,
                  nq, qtabs, qdmats, qorders, qnames, ltype, lorder, lsoln, maxstart);
    return 0;
}
