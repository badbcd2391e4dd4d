/*
 * oracle_cli.c - the reference's "cudaSaTabsearch -c" command line, restated
 * around the CPU oracle (TEST INFRASTRUCTURE, see sa_oracle.h).
 *
 * Follows nvcc_src_current/cudaSaTabsearch.cu: options :605-626, "-q" SID list
 * :631-664, stdin header :667-694, LTYPE override :696-700, db load :702-727,
 * SID lookup :730-784, srand48(1234) :871, small class then large class
 * :1276-1309, per-query body tabsearch_host_thread :308-459.
 *
 * Extra switches (not in the reference): -p selects the counter-based Philox
 * streams the GPU kernel uses (sa_oracle.h), -s SEED sets their seed, -S SEED
 * seeds the sequential drand48 stream (the reference hard-codes srand48(1234),
 * :871; other seeds measure its seed-to-seed spread,
 * tests/golden/make_seed_spread.py), -m N moves
 * the small/large class boundary (96 today, 32 in the 2013 snapshot whose
 * recorded output is one of the golden files).
 *
 * The ASCII reader and the Gumbel statistics are the product's host C files
 * (cuda_satabsearch_amd/csrc/host/); byte-identical stdout against oracle/_ref
 * pins them together with the search itself.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>
#include <time.h>

#include "sa_oracle.h"
#include "sat_parse.h"
#include "sat_gumbel.h"

typedef struct dense_class {
    int       count;
    int       pitch;
    int      *set_index;   /* index into the struct set */
    int      *orders;
    int64_t  *ordinal;     /* db file-order ordinal */
    uint8_t  *tabs;
    float    *dmats;
} dense_class;

static void build_class(const sat_struct_set *db, int lo, int hi, dense_class *c)
{
    memset(c, 0, sizeof(*c));
    for (int s = 0; s < db->count; s++)
        if (db->order[s] > lo && db->order[s] <= hi) {
            c->count++;
            if (db->order[s] > c->pitch) c->pitch = db->order[s];
        }
    if (c->pitch < 1) c->pitch = 1;
    size_t cells = (size_t)c->pitch * c->pitch;
    c->set_index = (int *)malloc(sizeof(int) * (c->count + 1));
    c->orders = (int *)malloc(sizeof(int) * (c->count + 1));
    c->ordinal = (int64_t *)malloc(sizeof(int64_t) * (c->count + 1));
    c->tabs = (uint8_t *)calloc(cells * (c->count + 1), 1);
    c->dmats = (float *)calloc(cells * (c->count + 1), sizeof(float));
    if (!c->set_index || !c->orders || !c->ordinal || !c->tabs || !c->dmats) {
        fprintf(stderr, "malloc db class failed\n");
        exit(1);
    }
    int d = 0;
    for (int s = 0; s < db->count; s++)
        if (db->order[s] > lo && db->order[s] <= hi) {
            c->set_index[d] = s;
            c->orders[d] = db->order[s];
            c->ordinal[d] = s;
            sat_set_expand(db, s, c->pitch, c->tabs + cells * d, c->dmats + cells * d);
            d++;
        }
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char *argv[])
{
    char dbfile[SAT_MAX_LINE_LEN] = "";
    char buf[SAT_MAX_LINE_LEN];
    int gpu_row_format = 0;   /* -G: large-class rows as the reference GPU path prints them (cudaSaTabsearch.cu:1261) */
    int querydbmode = 0, maxstart = 128, philox = 0, small_limit = SAT_MAXDIM_SMALL;
    unsigned long long seed = 1234;
    long drand_seed = 1234;
    int ltype = 0, lorder = 0, lsoln = 0;
    char cltype, clorder, clsoln;
    int c;

    while ((c = getopt(argc, argv, "cq:r:ps:S:m:tG")) != -1) {
        switch (c) {
        case 'c': break; /* always host */
        case 'q': querydbmode = 1; strncpy(dbfile, optarg, sizeof(dbfile) - 1); break;
        case 'r': maxstart = atoi(optarg); break;
        case 'p': philox = 1; break;
        case 's': seed = strtoull(optarg, NULL, 0); break;
        case 'S': drand_seed = strtol(optarg, NULL, 0); break;
        case 't': sa_oracle_trace = 1; break;
        case 'G': gpu_row_format = 1; break;
        case 'm': small_limit = atoi(optarg); break; /* 2013 snapshot: MAXDIM_GPU = 32 */
        default:
            fprintf(stderr, "Usage: %s [-c] [-q dbfile] [-r restarts] [-p] [-s seed] [-S drand48_seed] [-m small_class_limit]\n", argv[0]);
            exit(1);
        }
    }

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
            sid_list = (char *)realloc(sid_list, (size_t)(num_queries + 1) * (SAT_LABELSIZE + 1));
            char *sid = sid_list + (size_t)num_queries * (SAT_LABELSIZE + 1);
            memset(sid, 0, SAT_LABELSIZE + 1);
            strncpy(sid, buf, SAT_LABELSIZE);
            sid[SAT_LABELSIZE - 1] = '\0';             /* SIDs are cut to 7 chars */
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
        ltype = 1; cltype = 'T';
    }

    FILE *dbfp = fopen(dbfile, "r");
    if (!dbfp) {
        fprintf(stderr, "ERROR opening db file %s\n", dbfile);
        exit(1);
    }
    double t0 = now_ms();
    int total = sat_read_structures(dbfp, &db, "database");
    fclose(dbfp);
    if (total < 0) {
        fprintf(stderr, "ERROR loading database\n");
        exit(1);
    }
    dense_class cls[2];
    build_class(&db, -(1 << 30), small_limit, &cls[0]);
    build_class(&db, small_limit, SAT_MAXDIM, &cls[1]);
    fprintf(stderr, "Loaded %d db entries (%d order > %d) in %f ms\n",
            total, cls[1].count, small_limit, now_ms() - t0);

    /* -q: SID -> structure, small class searched first, then large (:746-780) */
    int *query_set_index = NULL;
    if (querydbmode) {
        query_set_index = (int *)malloc(sizeof(int) * (num_queries + 1));
        for (int i = 0; i < num_queries; i++) {
            const char *sid = sid_list + (size_t)i * (SAT_LABELSIZE + 1);
            int found = -1;
            for (int k = 0; k < 2 && found < 0; k++)
                for (int d = 0; d < cls[k].count; d++)
                    if (!strcasecmp(sid, sat_set_name(&db, cls[k].set_index[d]))) {
                        found = cls[k].set_index[d];
                        break;
                    }
            if (found < 0) {
                fprintf(stderr, "ERROR: query %s not found\n", sid);
                exit(1);
            }
            query_set_index[i] = found;
        }
    }

    fprintf(stderr, "maxstart = %d\n", maxstart);
    sa_oracle_rng rng;
    memset(&rng, 0, sizeof(rng));
    rng.mode = philox ? SA_RNG_PHILOX : SA_RNG_DRAND48;
    rng.lcg = sa_oracle_srand48(drand_seed);
    rng.seed = seed;

    uint8_t *qtab = (uint8_t *)calloc(SA_MAXDIM * SA_MAXDIM, 1);
    float *qdmat = (float *)calloc(SA_MAXDIM * SA_MAXDIM, sizeof(float));
    uint8_t qtypes[SA_MAXDIM];

    for (int k = 0; k < 2; k++) {
        dense_class *cl = &cls[k];
        if (k == 1 && cl->count == 0)
            break;
        int *scores = (int *)malloc(sizeof(int) * (cl->count + 1));
        int *ssemaps = (int *)malloc(sizeof(int) * SA_MAXDIM * ((size_t)cl->count + 1));
        for (int qi = 0; qi < num_queries; qi++) {
            const sat_struct_set *src = querydbmode ? &db : &queries;
            int s = querydbmode ? query_set_index[qi] : qi;
            sa_oracle_query q;
            q.n = src->order[s];
            q.pitch = SA_MAXDIM;
            sat_set_expand(src, s, SA_MAXDIM, qtab, qdmat);
            for (int i = 0; i < q.n; i++)
                qtypes[i] = qtab[i * SA_MAXDIM + i];
            q.tab = qtab;
            q.dmat = qdmat;
            q.ssetypes = qtypes;
            rng.query_ordinal = (uint32_t)qi;

            printf("# cudaSaTabsearch LTYPE = %c LORDER = %c LSOLN = %c\n",
                   ltype ? 'T' : 'F', lorder ? 'T' : 'F', lsoln ? 'T' : 'F');
            printf("# QUERY ID = %-8s\n", sat_set_name(src, s));
            printf("# DBFILE = %-80s\n", dbfile);

            double t1 = now_ms();
            sa_oracle_search(&q, cl->count, cl->orders, cl->ordinal, cl->tabs, cl->dmats,
                             cl->pitch, lorder, lsoln, maxstart, &rng, scores, ssemaps);
            double ms = now_ms() - t1;
            fprintf(stderr, "host execution time %f ms\n", ms);
            fprintf(stderr, "%f million iterations/sec\n",
                    ((double)cl->count * ((double)maxstart * SA_MAXITER) / (ms / 1000)) / 1.0e6);

            for (int d = 0; d < cl->count; d++) {
                double norm2score = sat_norm2(scores[d], q.n, cl->orders[d]);
                double zscore = sat_z_gumbel_trunc(norm2score);
                double pvalue = sat_pv_gumbel(zscore);
                printf(gpu_row_format && k == 1 ? "%-8s %d %g %g  %g\n" : "%-8s %d %g %g %g\n",
                       sat_set_name(&db, cl->set_index[d]), scores[d], norm2score, zscore, pvalue);
                if (lsoln)
                    for (int i = 0; i < q.n; i++)
                        if (ssemaps[(size_t)d * SA_MAXDIM + i] >= 0)
                            printf("%3d %3d\n", i + 1, ssemaps[(size_t)d * SA_MAXDIM + i] + 1);
            }
        }
        free(scores);
        free(ssemaps);
    }
    (void)cltype; (void)clorder; (void)clsoln;
    return 0;
}
