/*
 * ldpc_oracle.c -- CPU restatement (plain C, fp64, single thread) of the reference decoders.
 *
 * TEST INFRASTRUCTURE ONLY -- see ldpc_oracle.h.  Compiled with -ffp-contract=off so that
 * `y + s*alpha` stays two roundings, as in the reference build (Makefile:20, x86-64 SSE2, no FMA).
 *
 * The reference walks dense rh x nh block loops and physically rotates M-vectors with memcpy
 * (decoders.cpp:327-346).  This restatement keeps an explicit list of the non-empty circulants
 * ("edge blocks", row-major order = the reference's j-then-k loop order) and indexes the rotated
 * position directly: out[n] = in[(n+shift) mod M].
 */
#include "ldpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_VAL 32767.0 /* decoders.cpp:4299-4301: (1L<<15)-1 */

struct orc_code {
    int rh, nh, M, N, R;
    int ne;         /* non-empty circulants */
    int *e_row;     /* [ne] block row    */
    int *e_col;     /* [ne] block column */
    int *e_shift;   /* [ne] circulant shift, reduced into [0,M) like rotate() does (:335-339) */
    int *row_start; /* [rh+1] edge-block range of each block row (columns ascending) */
    int *col_start; /* [nh+1] */
    int *col_edge;  /* [ne]  edge-block ids of each block column, rows ascending */
    /* per-call workspace */
    double *soft;   /* [N] */
    double *min1, *min2; /* [R] */
    int *pos;       /* [R] */
    unsigned char *sgn; /* [R] */
    double *nmin1, *nmin2; int *npos; unsigned char *nsgn; /* [M] row under construction */
    unsigned char *S;   /* [ne*M] per-edge v2c sign (reference: dense short ms_BnNS[rh*N], decoders.h:217) */
    unsigned char *synd; /* [R] */
    double *tmp;    /* [max_row_weight*M] */
    /* sum-product */
    double *yd, *s;  /* [N], [R] */
    double *ZZ;      /* [ne*M]  message of edge-block e at VARIABLE position t (reference: ZZ[j][k*M+t]) */
    double *ZZ0;     /* [rh*M] scratch: ZZ0[j][t] */
    unsigned char *bp_BB;    /* [ne*M] */
    unsigned char *bp_bs;    /* [R] */
    unsigned char *bp_syndr; /* [R] PERSISTS between calls, like DEC_STATE::syndr */
    /* integer min-sum */
    short *isoft, *iy, *imin1, *imin2, *inmin1, *inmin2;
};

static int mod_shift(int c, int M) {
    while (c < 0) c += M;
    while (c >= M) c -= M;
    return c;
}

orc_code *orc_open(int rh, int nh, int M, const short *hd) {
    orc_code *c = (orc_code *)calloc(1, sizeof(*c));
    int j, k, e, maxw = 0;
    if (!c) return NULL;
    c->rh = rh; c->nh = nh; c->M = M; c->N = nh * M; c->R = rh * M;
    for (j = 0; j < rh * nh; j++) if (hd[j] != -1) c->ne++;
    c->e_row = (int *)malloc(sizeof(int) * (c->ne + 1));
    c->e_col = (int *)malloc(sizeof(int) * (c->ne + 1));
    c->e_shift = (int *)malloc(sizeof(int) * (c->ne + 1));
    c->row_start = (int *)malloc(sizeof(int) * (rh + 1));
    c->col_start = (int *)malloc(sizeof(int) * (nh + 1));
    c->col_edge = (int *)malloc(sizeof(int) * (c->ne + 1));
    e = 0;
    for (j = 0; j < rh; j++) {
        c->row_start[j] = e;
        for (k = 0; k < nh; k++) {
            if (hd[j * nh + k] != -1) {
                c->e_row[e] = j; c->e_col[e] = k; c->e_shift[e] = mod_shift(hd[j * nh + k], M);
                e++;
            }
        }
        if (e - c->row_start[j] > maxw) maxw = e - c->row_start[j];
    }
    c->row_start[rh] = e;
    e = 0;
    for (k = 0; k < nh; k++) {
        int i;
        c->col_start[k] = e;
        for (i = 0; i < c->ne; i++) if (c->e_col[i] == k) c->col_edge[e++] = i; /* rows ascending */
    }
    c->col_start[nh] = e;

    c->soft = (double *)calloc(c->N, sizeof(double));
    c->min1 = (double *)calloc(c->R, sizeof(double));
    c->min2 = (double *)calloc(c->R, sizeof(double));
    c->pos = (int *)calloc(c->R, sizeof(int));
    c->sgn = (unsigned char *)calloc(c->R, 1);
    c->nmin1 = (double *)calloc(M, sizeof(double));
    c->nmin2 = (double *)calloc(M, sizeof(double));
    c->npos = (int *)calloc(M, sizeof(int));
    c->nsgn = (unsigned char *)calloc(M, 1);
    c->S = (unsigned char *)calloc((size_t)(c->ne + 1) * M, 1);
    c->synd = (unsigned char *)calloc(c->R, 1);
    c->tmp = (double *)calloc((size_t)(maxw + 1) * M, sizeof(double));
    c->yd = (double *)calloc(c->N, sizeof(double));
    c->s = (double *)calloc(c->R, sizeof(double));
    c->ZZ = (double *)calloc((size_t)(c->ne + 1) * M, sizeof(double));
    c->ZZ0 = (double *)calloc((size_t)rh * M, sizeof(double));
    c->bp_BB = (unsigned char *)calloc((size_t)(c->ne + 1) * M, 1);
    c->bp_bs = (unsigned char *)calloc(c->R, 1);
    c->bp_syndr = (unsigned char *)calloc(c->R, 1);
    c->isoft = (short *)calloc(c->N, sizeof(short));
    c->iy = (short *)calloc(c->N, sizeof(short));
    c->imin1 = (short *)calloc(c->R, sizeof(short));
    c->imin2 = (short *)calloc(c->R, sizeof(short));
    c->inmin1 = (short *)calloc(M, sizeof(short));
    c->inmin2 = (short *)calloc(M, sizeof(short));
    return c;
}

void orc_close(orc_code *c) {
    if (!c) return;
    free(c->e_row); free(c->e_col); free(c->e_shift); free(c->row_start); free(c->col_start); free(c->col_edge);
    free(c->soft); free(c->min1); free(c->min2); free(c->pos); free(c->sgn);
    free(c->nmin1); free(c->nmin2); free(c->npos); free(c->nsgn); free(c->S); free(c->synd); free(c->tmp);
    free(c->yd); free(c->s); free(c->ZZ); free(c->ZZ0);
    free(c->bp_BB); free(c->bp_bs); free(c->bp_syndr);
    free(c->isoft); free(c->iy); free(c->imin1); free(c->imin2); free(c->inmin1); free(c->inmin2);
    free(c);
}

int orc_n(const orc_code *c) { return c->N; }
int orc_r(const orc_code *c) { return c->R; }
int orc_edges(const orc_code *c) { return c->ne; }

/* rotated position of check-lane n in a circulant of shift sh */
#define ROT(n, sh, M) (((n) + (sh)) < (M) ? ((n) + (sh)) : ((n) + (sh) - (M)))

/* decoders.cpp:793-814 */
int orc_syndrome_nonzero(const orc_code *c, const double *soft) {
    int e, n, M = c->M, parity = 0;
    memset(c->synd, 0, c->R);
    for (e = 0; e < c->ne; e++) {
        const double *col = soft + c->e_col[e] * M;
        unsigned char *sy = c->synd + c->e_row[e] * M;
        int sh = c->e_shift[e];
        for (n = 0; n < M; n++) sy[n] ^= (col[ROT(n, sh, M)] < 0);
    }
    for (n = 0; n < c->R; n++) parity |= c->synd[n];
    return parity;
}

/* ---------------------------------------------------------------------------------------------
 * Flooding normalised min-sum: decoders.cpp:4554-4767
 * ------------------------------------------------------------------------------------------- */
int orc_min_sum(orc_code *c, const double *y, double *decword, int maxsteps, int decision, double alpha) {
    const int M = c->M, N = c->N, R = c->R, rh = c->rh;
    int iter, j, e, n, v, parity = 0;
    double *soft = c->soft;

    /* :4579-4596 records and edge signs start at zero */
    memset(c->min1, 0, sizeof(double) * R);
    memset(c->min2, 0, sizeof(double) * R);
    memset(c->pos, 0, sizeof(int) * R);
    memset(c->sgn, 0, R);
    memset(c->S, 0, (size_t)c->ne * M);
    /* :4602-4622 initial syndrome of y: its value is never consulted (the loop below recomputes parity
     * before every test), except when maxsteps <= 0 -- then the reference returns on that parity. */
    if (maxsteps <= 0) {
        /* the reference XORs into a possibly stale syndr[]; with a fresh state it is the plain syndrome */
        parity = orc_syndrome_nonzero(c, y);
        return parity ? 0 : 1; /* iter == 0: parity ? -0 : 0+1 */
    }

    for (iter = 0; iter < maxsteps; iter++) {
        /* STATE 1 (:4637-4667): soft[v] = sum over the column's blocks, ascending block row, from 0.0 */
        memset(soft, 0, sizeof(double) * N);
        for (e = 0; e < c->ne; e++) {
            const int jj = c->e_row[e], k = c->e_col[e], sh = c->e_shift[e];
            const double *m1 = c->min1 + jj * M, *m2 = c->min2 + jj * M;
            const int *ps = c->pos + jj * M;
            const unsigned char *sg = c->sgn + jj * M, *S = c->S + (size_t)e * M;
            double *col = soft + k * M;
            for (n = 0; n < M; n++) {
                double a = (ps[n] == k) ? m2[n] : m1[n];
                double m = (S[n] ^ sg[n]) ? -a : a;
                int t = ROT(n, sh, M);
                col[t] = col[t] + m;
            }
        }
        /* STATE 2 (:4670-4685): two roundings, multiply then add */
        for (v = 0; v < N; v++) {
            double p = soft[v] * alpha;
            soft[v] = y[v] + p;
            decword[v] = decision ? soft[v] : (double)(soft[v] < 0);
        }
        /* STATE 3 (:4690-4755) */
        memset(c->synd, 0, R);
        for (j = 0; j < rh; j++) {
            double *m1 = c->min1 + j * M, *m2 = c->min2 + j * M;
            int *ps = c->pos + j * M;
            unsigned char *sg = c->sgn + j * M, *sy = c->synd + j * M;
            for (n = 0; n < M; n++) { c->nmin1[n] = ORC_MAX_VAL; c->nmin2[n] = ORC_MAX_VAL; c->npos[n] = 0; c->nsgn[n] = 0; }
            for (e = c->row_start[j]; e < c->row_start[j + 1]; e++) {
                const int k = c->e_col[e], sh = c->e_shift[e];
                const double *col = soft + k * M;
                unsigned char *S = c->S + (size_t)e * M;
                for (n = 0; n < M; n++) {
                    double r = col[ROT(n, sh, M)];
                    double a, t, val;
                    unsigned char sn;
                    sy[n] ^= (r < 0);
                    a = ((ps[n] == k) ? m2[n] : m1[n]) * alpha;
                    t = (S[n] ^ sg[n]) ? -a : a;
                    t = r - t;
                    sn = (t < 0);
                    S[n] = sn;
                    c->nsgn[n] ^= sn;
                    val = t < 0.0 ? -t : t;
                    val = (val > ORC_MAX_VAL) ? ORC_MAX_VAL : val;
                    if (val < c->nmin1[n]) { c->npos[n] = k; c->nmin2[n] = c->nmin1[n]; c->nmin1[n] = val; }
                    else if (val < c->nmin2[n]) c->nmin2[n] = val;
                }
            }
            for (n = 0; n < M; n++) { m1[n] = c->nmin1[n]; m2[n] = c->nmin2[n]; ps[n] = c->npos[n]; sg[n] = c->nsgn[n]; }
        }
        parity = 0;
        for (n = 0; n < R; n++) parity |= c->synd[n];
        if (!parity) break;
    }
    return parity ? -iter : iter + 1; /* :4766 */
}

/* ---------------------------------------------------------------------------------------------
 * Layered offset min-sum: decoders.cpp:5064-5425, active branch :5106-5290 (MY_VERSION :5140-5207)
 * ------------------------------------------------------------------------------------------- */
int orc_lmin_sum(orc_code *c, const double *y, double *decword, int maxsteps, int decision) {
    const int M = c->M, N = c->N, R = c->R, rh = c->rh;
    const double beta = 0.4; /* :5163 */
    int iter, j, e, n, v, parity;
    double *soft = c->soft;

    memcpy(soft, y, sizeof(double) * N);         /* :5088 */
    memset(c->min1, 0, sizeof(double) * R);      /* :5091-5097 */
    memset(c->min2, 0, sizeof(double) * R);
    memset(c->pos, 0, sizeof(int) * R);
    memset(c->sgn, 0, R);
    memset(c->S, 0, (size_t)c->ne * M);          /* :5108 */
    parity = orc_syndrome_nonzero(c, soft);       /* :5111-5115 */

    for (iter = 0; iter < maxsteps; iter++) {
        if (!parity) break;                       /* :5119 */
        for (j = 0; j < rh; j++) {                /* layers are strictly sequential */
            double *m1 = c->min1 + j * M, *m2 = c->min2 + j * M;
            int *ps = c->pos + j * M;
            unsigned char *sg = c->sgn + j * M;
            const int e0 = c->row_start[j], e1 = c->row_start[j + 1];
            for (n = 0; n < M; n++) { c->nmin1[n] = ORC_MAX_VAL; c->nmin2[n] = ORC_MAX_VAL; c->npos[n] = 0; c->nsgn[n] = 0; }
            for (e = e0; e < e1; e++) {           /* :5141-5177 */
                const int k = c->e_col[e], sh = c->e_shift[e];
                const double *col = soft + k * M;
                unsigned char *S = c->S + (size_t)e * M;
                double *tv = c->tmp + (size_t)(e - e0) * M; /* reference parks this in soft[k*M+n] (:5170) */
                for (n = 0; n < M; n++) {
                    double r = col[ROT(n, sh, M)];
                    double a = (ps[n] == k) ? m2[n] : m1[n];
                    double pc = (S[n] ^ sg[n]) ? -a : a;
                    double t = r - pc;
                    unsigned char sn = (t < 0);
                    double mag = (t < 0.0 ? -t : t);
                    mag -= beta;
                    mag = mag < 0 ? 0 : mag;
                    tv[n] = t;
                    S[n] = sn;
                    /* process_check_node :5012-5027 (no MAX_VAL clamp in this decoder) */
                    c->nsgn[n] ^= sn;
                    if (mag < c->nmin1[n]) { c->npos[n] = k; c->nmin2[n] = c->nmin1[n]; c->nmin1[n] = mag; }
                    else if (mag < c->nmin2[n]) c->nmin2[n] = mag;
                }
            }
            for (n = 0; n < M; n++) { m1[n] = c->nmin1[n]; m2[n] = c->nmin2[n]; ps[n] = c->npos[n]; sg[n] = c->nsgn[n]; }
            for (e = e0; e < e1; e++) {           /* :5182-5207 */
                const int k = c->e_col[e], sh = c->e_shift[e];
                double *col = soft + k * M;
                const unsigned char *S = c->S + (size_t)e * M;
                const double *tv = c->tmp + (size_t)(e - e0) * M;
                for (n = 0; n < M; n++) {
                    double a = (ps[n] == k) ? m2[n] : m1[n];
                    double cv = (S[n] ^ sg[n]) ? -a : a;
                    col[ROT(n, sh, M)] = tv[n] + cv;
                }
            }
        }
        parity = orc_syndrome_nonzero(c, soft);   /* :5281-5284 */
        if (!parity) break;                       /* :5287 */
    }
    for (v = 0; v < N; v++) decword[v] = decision ? soft[v] : (double)(soft[v] < 0); /* :5413-5422 */
    return parity ? -iter : iter + 1;             /* :5424 */
}

/* ---------------------------------------------------------------------------------------------
 * Sum-product in the likelihood-ratio domain: decoders.cpp:1923-2185
 * ------------------------------------------------------------------------------------------- */
static double orc_mind(double a, double b) { if (a < b) return a; else return b; } /* :104 */
static double orc_maxd(double a, double b) { if (a < b) return b; else return a; } /* :105 */

int orc_sum_prod(orc_code *c, double *soft, double *decword, int maxiter, int decision) {
    const int M = c->M, N = c->N, R = c->R, nh = c->nh;
    int i, e, n, t, v, q, synd, iter = 0;
    double *yd = c->yd, *s = c->s;

    for (v = 0; v < N; v++) {                     /* :1947-1951, INPUT_LIMIT 20.0 (:94) */
        double yl = orc_maxd(orc_mind(soft[v], 20.0), -20.0);
        yd[v] = soft[v] = exp(yl);
    }
    for (n = 0; n < R; n++) s[n] = 0;             /* :1954 */
    for (e = 0; e < c->ne; e++)                   /* :1957-1959 (dense rh x N in the reference) */
        for (t = 0; t < M; t++) c->ZZ[(size_t)e * M + t] = 1.0;

    /* :1964-2002 input check, SP_THR = 1.0 */
    memset(c->synd, 0, R);
    for (e = 0; e < c->ne; e++) {
        const double *col = soft + c->e_col[e] * M;
        unsigned char *sy = c->synd + c->e_row[e] * M;
        int sh = c->e_shift[e];
        for (n = 0; n < M; n++) sy[n] ^= (col[ROT(n, sh, M)] < 1.0);
    }
    synd = 0;
    for (n = 0; n < R; n++) synd |= c->synd[n];
    if (!synd) {
        for (v = 0; v < N; v++) decword[v] = decision ? soft[v] : (double)(soft[v] < 1.0);
        return 0;
    }

    while (iter < maxiter) {
        memset(c->synd, 0, R);                    /* :2009-2011 */
        for (n = 0; n < R; n++) s[n] = 1.0;
        for (v = 0; v < N; v++) soft[v] = yd[v];

        for (i = 0; i < nh; i++) {                /* :2013-2060 pass 1, block column by block column */
            const int q0 = c->col_start[i], q1 = c->col_start[i + 1];
            for (q = q0; q < q1; q++) {
                const int ee = c->col_edge[q], j = c->e_row[ee], sh = c->e_shift[ee];
                double *z0 = c->ZZ0 + (size_t)j * M;
                double *sj = s + j * M;
                int qq;
                for (t = 0; t < M; t++) {
                    double AA = yd[i * M + t];    /* :2027 */
                    for (qq = q0; qq < q1; qq++) { /* ascending block row, skipping this one (:2029-2041) */
                        if (qq == q) continue;
                        AA *= c->ZZ[(size_t)c->col_edge[qq] * M + t];
                    }
                    z0[t] = (AA - 1) / (AA + 1);  /* :2044 */
                }
                for (n = 0; n < M; n++) sj[n] *= z0[ROT(n, sh, M)]; /* :2047-2050 */
            }
            for (q = q0; q < q1; q++) {           /* :2054-2060 */
                const int ee = c->col_edge[q], j = c->e_row[ee];
                memcpy(c->ZZ + (size_t)ee * M, c->ZZ0 + (size_t)j * M, sizeof(double) * M);
            }
        }

        for (i = 0; i < nh; i++) {                /* :2103-2144 pass 2 */
            const int q0 = c->col_start[i], q1 = c->col_start[i + 1];
            for (q = q0; q < q1; q++) {
                const int ee = c->col_edge[q], j = c->e_row[ee], sh = c->e_shift[ee];
                const double *sj = s + j * M;
                double *zz = c->ZZ + (size_t)ee * M;
                for (t = 0; t < M; t++) {
                    int nn = t - sh; if (nn < 0) nn += M;       /* rotate by M-circ (:2113) */
                    double A = sj[nn] / zz[t];
                    A = (1 + A) / (1 - A);
                    A = orc_maxd(orc_mind(A, 1.9e+8), -5.2e-9); /* :2120 (sic: negative lower clamp) */
                    zz[t] = A;
                    soft[i * M + t] *= A;
                }
            }
            for (q = q0; q < q1; q++) {           /* :2129-2142 */
                const int ee = c->col_edge[q], j = c->e_row[ee], sh = c->e_shift[ee];
                unsigned char *sy = c->synd + j * M;
                const double *col = soft + i * M;
                for (n = 0; n < M; n++) sy[n] ^= (col[ROT(n, sh, M)] < 1.0);
            }
        }

        synd = 0;
        for (n = 0; n < R; n++) synd |= c->synd[n];
        if (!synd) {                              /* :2151-2168 */
            for (v = 0; v < N; v++) decword[v] = decision ? soft[v] : (double)(soft[v] < 1.0);
            iter++;
            return iter;
        }
        iter++;
    }
    for (v = 0; v < N; v++) decword[v] = decision ? soft[v] : (double)(soft[v] < 1.0);
    return -iter;                                 /* :2184 */
}


/* ---------------------------------------------------------------------------------------------
 * TDMP (layered) sum-product in the probability domain: decoders.cpp:2584-2744, map_bin :2191-2228,
 * check_syndrome_thr :2274-2306.
 * ------------------------------------------------------------------------------------------- */
static int orc_syndrome_thr(const orc_code *c, const double *p, double thr) {
    int e, n, M = c->M, parity = 0;
    memset(c->synd, 0, c->R);
    for (e = 0; e < c->ne; e++) {
        const double *col = p + c->e_col[e] * M;
        unsigned char *sy = c->synd + c->e_row[e] * M;
        int sh = c->e_shift[e];
        for (n = 0; n < M; n++) sy[n] ^= (col[ROT(n, sh, M)] > thr);
    }
    for (n = 0; n < c->R; n++) parity |= c->synd[n];
    return parity;
}

/* :2191-2228 with step == 1; rw >= 2 */
static void orc_map_bin(double *a, int rw, double *P, double *SF, double *SB) {
    int i;
    for (i = 0; i < rw; i++) P[i] = 1 - 2 * a[i];
    SB[0] = 1.0; SF[rw - 1] = 0.0;
    SF[0] = P[0];
    for (i = 1; i < rw - 1; i++) SF[i] = P[i] * SF[i - 1];
    SB[rw - 1] = P[rw - 1];
    for (i = rw - 2; i > 0; i--) SB[i] = P[i] * SB[i + 1];
    a[0] = (1 - SB[1]) / 2;
    for (i = 1; i < rw - 1; i++) { double Z = SF[i - 1] * SB[i + 1]; a[i] = (1 - Z) / 2; }
    a[rw - 1] = (1 - SF[rw - 2]) / 2;
}

int orc_tdmp_sum_prod(orc_code *c, double *soft, double *decword, int maxsteps, double *post_out) {
    const int M = c->M, N = c->N, rh = c->rh;
    const double T = 0.0001, TT = 0; /* :2597-2598 */
    double *so = c->soft;             /* soft_out */
    double *Zs = c->ZZ;               /* per-edge state, [edge block][check n] (reference: Z[check][cnt]) */
    double y[64], a[64], P[64], SF[64], SB[64];
    int v, e, j, n, steps, synd;

    for (v = 0; v < N; v++) {         /* :2611-2618 */
        double x = soft[v] * 0.5;
        double yy = orc_maxd(orc_mind(x, 20.0), -20.0);
        double e0 = exp(yy), e1 = exp(-yy);
        soft[v] = e1 / (e0 + e1);
    }
    for (e = 0; e < c->ne; e++) for (n = 0; n < M; n++) Zs[(size_t)e * M + n] = 0.5;  /* :2620-2641 */
    for (v = 0; v < N; v++) so[v] = soft[v];
    synd = orc_syndrome_thr(c, so, 0.5);  /* :2653 */
    if (synd == 0) {
        for (v = 0; v < N; v++) decword[v] = so[v] > 0.5;
        if (post_out) memcpy(post_out, so, sizeof(double) * N);
        return 0;
    }
    steps = 0;
    while (steps < maxsteps) {
        for (j = 0; j < rh; j++) {    /* layers, :2668-2724 */
            const int e0 = c->row_start[j], rw = c->row_start[j + 1] - e0;
            for (n = 0; n < M; n++) {
                int s;
                for (s = 0; s < rw; s++) {
                    const int idx = c->e_col[e0 + s] * M + ROT(n, c->e_shift[e0 + s], M);
                    const double x = so[idx], aa = Zs[(size_t)(e0 + s) * M + n];
                    y[s] = x * (1.0 - aa) / (aa + x - 2.0 * aa * x);          /* :2686 */
                }
                for (s = 0; s < rw; s++) {
                    if (y[s] < TT) y[s] = TT;
                    if (y[s] > 1 - TT) y[s] = 1 - TT;
                    a[s] = y[s];
                }
                orc_map_bin(a, rw, P, SF, SB);
                for (s = 0; s < rw; s++) {
                    if (a[s] < T) a[s] = T;
                    if (a[s] > 1.0 - T) a[s] = 1.0 - T;
                }
                for (s = 0; s < rw; s++) {
                    const int idx = c->e_col[e0 + s] * M + ROT(n, c->e_shift[e0 + s], M);
                    Zs[(size_t)(e0 + s) * M + n] = a[s];
                    so[idx] = y[s] * a[s] / (1.0 - y[s] - a[s] + 2 * y[s] * a[s]);  /* :2716 */
                }
            }
        }
        synd = orc_syndrome_thr(c, so, 0.5);  /* the reference recomputes it after every layer; only the last one counts (:2723) */
        steps = steps + 1;
        if (synd == 0) break;
    }
    for (v = 0; v < N; v++) decword[v] = so[v] > 0.5;  /* :2737-2738 */
    if (post_out) memcpy(post_out, so, sizeof(double) * N);
    if (synd == 1) steps = -steps;
    return steps;
}

/* ---------------------------------------------------------------------------------------------
 * Flooding sum-product in the probability domain ("advanced sum-product"): decoders.cpp:2324-2581
 * ------------------------------------------------------------------------------------------- */
int orc_sum_prod_gf2(orc_code *c, double *soft, double *decword, int maxsteps, int decision) {
    const int M = c->M, N = c->N, rh = c->rh, nh = c->nh;
    double *so = c->soft;   /* soft_out */
    double *st = c->ZZ;     /* state[edge block][CHECK position n] (reference: state[slot][j*M+n]) */
    double a[64], P[64], SF[64], SB[64];
    int v, e, j, n, i, q, steps, synd, all2 = 1;

    for (i = 0; i < nh; i++) if (c->col_start[i + 1] - c->col_start[i] != 2) all2 = 0; /* decod_init :1027-1044: asp_all_cw_2 */

    for (v = 0; v < N; v++) { /* :2351-2358 */
        double x = soft[v] * 0.5;
        double y = orc_maxd(orc_mind(x, 20.0), -20.0);
        double e0 = exp(y), e1 = exp(-y);
        soft[v] = e1 / (e0 + e1);
    }
    for (e = 0; e < c->ne; e++) { /* :2361-2379 */
        const double *col = soft + c->e_col[e] * M;
        for (n = 0; n < M; n++) st[(size_t)e * M + n] = col[ROT(n, c->e_shift[e], M)];
    }
    for (v = 0; v < N; v++) so[v] = soft[v];
    synd = orc_syndrome_thr(c, so, 0.5); /* :2393 */
    if (synd == 0) {
        for (v = 0; v < N; v++) decword[v] = decision ? so[v] : (double)(so[v] > 0.5);
        return 0;
    }
    steps = 0;
    while (steps < maxsteps) {
        for (j = 0; j < rh; j++) { /* check nodes :2406-2428 */
            const int e0 = c->row_start[j], rw = c->row_start[j + 1] - e0;
            for (n = 0; n < M; n++) {
                int s;
                for (s = 0; s < rw; s++) a[s] = st[(size_t)(e0 + s) * M + n];
                orc_map_bin(a, rw, P, SF, SB);
                for (s = 0; s < rw; s++) st[(size_t)(e0 + s) * M + n] = a[s];
            }
        }
        if (all2) { /* every block column has exactly two circulants: upstream's own branch :2431-2480, no clamping, messages formed
                     * from the channel value and the OTHER edge directly (hci[i][0] < hci[i][1]: rows ascending) */
            for (i = 0; i < nh; i++) {
                const int ea = c->col_edge[c->col_start[i]], eb = c->col_edge[c->col_start[i] + 1];
                for (n = 0; n < M; n++) {
                    int na = n - c->e_shift[ea], nb = n - c->e_shift[eb];
                    double p1, p0, q10, q11, q00, q01, d0, d1;
                    if (na < 0) na += M;
                    if (nb < 0) nb += M;
                    d0 = st[(size_t)ea * M + na]; /* data0[k], :2449 */
                    d1 = st[(size_t)eb * M + nb]; /* data1[k], :2450 */
                    p1 = soft[i * M + n];
                    q10 = p1; q11 = p1; p0 = 1.0 - p1; q00 = 1.0 - p1; q01 = 1.0 - p1; /* :2454-2459 */
                    q10 = q10 * d1;        /* :2461-2466 */
                    q00 = q00 * (1 - d1);
                    q11 = q11 * d0;
                    q01 = q01 * (1 - d0);
                    p1 = q10 * d0;
                    p0 = q00 * (1 - d0);
                    so[i * M + n] = p1 / (p0 + p1);            /* :2469 */
                    st[(size_t)ea * M + na] = q10 / (q10 + q00); /* :2471 */
                    st[(size_t)eb * M + nb] = q11 / (q11 + q01); /* :2472 */
                }
            }
        } else {
        for (i = 0; i < nh; i++) { /* symbol nodes, overall products :2488-2520 */
            for (n = 0; n < M; n++) {
                double P1 = soft[i * M + n], P0 = 1 - soft[i * M + n];
                for (q = c->col_start[i]; q < c->col_start[i + 1]; q++) {
                    const int ee = c->col_edge[q];
                    int nn = n - c->e_shift[ee]; if (nn < 0) nn += M; /* rotate by m-circ */
                    const double d = st[(size_t)ee * M + nn];
                    P1 *= d;
                    P0 *= 1 - d;
                }
                so[i * M + n] = P1 / (P0 + P1);
            }
        }
        for (i = 0; i < nh; i++) { /* local data updating :2523-2556 */
            for (q = c->col_start[i]; q < c->col_start[i + 1]; q++) {
                const int ee = c->col_edge[q];
                for (n = 0; n < M; n++) {
                    int nn = n - c->e_shift[ee]; if (nn < 0) nn += M;
                    const double sov = so[i * M + n], sos = st[(size_t)ee * M + nn];
                    const double p1 = sov / sos;
                    const double p0 = (1 - sov) / (1 - sos);
                    const double d = p1 / (p1 + p0);
                    st[(size_t)ee * M + nn] = orc_maxd(orc_mind(d, 1.0 - 0.000001), 0.000001); /* SP_DEC_MAX/MIN_VAL :96-97 */
                }
            }
        }
        }
        synd = orc_syndrome_thr(c, so, 0.5); /* :2566 */
        if (synd == 0) {
            for (v = 0; v < N; v++) decword[v] = decision ? so[v] : (double)(so[v] > 0.5);
            return steps + 1;
        }
        steps = steps + 1;
    }
    for (v = 0; v < N; v++) decword[v] = decision ? so[v] : (double)(so[v] > 0.5);
    return -steps;
}


/* ---------------------------------------------------------------------------------------------
 * Gallager belief propagation in the log domain: decoders.cpp:1708-1920 (BP_DEC, id 0; BP_USE_EPS is
 * not defined, decoders.h:9).  ZZ[e][t] / BB[e][t] are indexed by VARIABLE position like upstream's
 * ZZ[j][k*M+t].  The syndrome array is NOT cleared before the input check (:1742-1762): it still holds
 * the final syndrome of the previous call on this state, which is non-zero after a failed frame
 *  -- restated, with orc_bp_stale() to read / set that carried state.
 * ------------------------------------------------------------------------------------------- */
#ifndef ORC_BP_EXP
#define ORC_BP_EXP exp
#define ORC_BP_LOG log
#endif
unsigned char *orc_bp_stale(orc_code *c) { return c->bp_syndr; }

int orc_bp(orc_code *c, double *soft, double *decword, int maxiter, int decision) {
    const int M = c->M, N = c->N, R = c->R, nh = c->nh;
    double *yd = c->yd, *s = c->s, *ZZ = c->ZZ;
    unsigned char *BB = c->bp_BB, *bs = c->bp_bs, *sy = c->bp_syndr;
    int v, e, n, i, q, iter = 0, synd;

    memset(ZZ, 0, sizeof(double) * (size_t)c->ne * M);                          /* :1731-1733 */
    for (v = 0; v < N; v++) yd[v] = soft[v] = orc_maxd(orc_mind(soft[v], 20.0), -20.0); /* :1737-1738 */
    for (e = 0; e < c->ne; e++) {                                                /* :1742-1762, no memset first */
        const double *col = soft + c->e_col[e] * M;
        unsigned char *row = sy + c->e_row[e] * M;
        for (n = 0; n < M; n++) row[n] ^= (col[ROT(n, c->e_shift[e], M)] < 0);
    }
    synd = 0;
    for (n = 0; n < R; n++) synd |= sy[n];
    if (!synd) {
        for (v = 0; v < N; v++) decword[v] = decision ? soft[v] : (double)(soft[v] < 0);
        return 0;
    }
    while (iter < maxiter) {
        memset(sy, 0, R); memset(bs, 0, R); memset(s, 0, sizeof(double) * R);   /* :1788-1790 */
        for (i = 0; i < nh; i++) {                                               /* :1792-1829 columns outer, rows inner */
            for (q = c->col_start[i]; q < c->col_start[i + 1]; q++) {
                const int ee = c->col_edge[q], j = c->e_row[ee], sh = c->e_shift[ee];
                double *z = ZZ + (size_t)ee * M;
                unsigned char *b = BB + (size_t)ee * M;
                for (n = 0; n < M; n++) {
                    const double A = ORC_BP_EXP(soft[i * M + n] - z[n]);
                    const double x = ORC_BP_LOG(fabs((A - 1) / (A + 1)));
                    b[n] = A < 1;
                    z[n] = x;
                }
                for (n = 0; n < M; n++) {
                    const int t = ROT(n, sh, M);
                    s[j * M + n] += z[t];
                    bs[j * M + n] ^= b[t];
                }
            }
        }
        memcpy(soft, yd, sizeof(double) * N);                                    /* :1834 */
        for (i = 0; i < nh; i++) {                                               /* :1836-1866 */
            for (q = c->col_start[i]; q < c->col_start[i + 1]; q++) {
                const int ee = c->col_edge[q], j = c->e_row[ee], sh = c->e_shift[ee];
                double *z = ZZ + (size_t)ee * M;
                const unsigned char *b = BB + (size_t)ee * M;
                for (n = 0; n < M; n++) {
                    int nn = n - sh; if (nn < 0) nn += M;                         /* rotate by m - circ */
                    double A = ORC_BP_EXP(s[j * M + nn] - z[n]);
                    const int bb = bs[j * M + nn] ^ b[n];
                    A = (1 - 2 * bb) * ORC_BP_LOG((1 + A) / (1 - A));
                    z[n] = orc_maxd(orc_mind(A, 19.07), -19.07);
                }
                for (n = 0; n < M; n++) soft[i * M + n] += z[n];
            }
        }
        for (e = 0; e < c->ne; e++) {                                            /* :1869-1887 (sy was cleared above) */
            const double *col = soft + c->e_col[e] * M;
            unsigned char *row = sy + c->e_row[e] * M;
            for (n = 0; n < M; n++) row[n] ^= (col[ROT(n, c->e_shift[e], M)] < 0);
        }
        synd = 0;
        for (n = 0; n < R; n++) synd |= sy[n];
        iter++;
        if (!synd) break;
    }
    for (v = 0; v < N; v++) decword[v] = decision ? soft[v] : (double)(soft[v] < 0);
    return synd ? -iter : iter;                                                  /* :1901-1919 */
}

/* ---------------------------------------------------------------------------------------------
 * Integer min-sum: decoders.cpp:5430-5690 (MS_MUL_CORRECTION, MS_ALPHA_FPP = 4, decoders.h:13-14)
 * ------------------------------------------------------------------------------------------- */
static short orc_limit(int x, short mx) { return (short)(x > mx ? mx : (x < -mx ? -mx : x)); } /* :4308 */

int orc_imin_sum(orc_code *c, const double *y, double *decword, int maxsteps, int decision,
                 double alpha, double thr, int qbits, int dbits) {
    const int M = c->M, N = c->N, R = c->R, rh = c->rh;
    const short max_data = (short)((1L << (dbits - 1)) - 1);  /* :5445 */
    const short max_quant = (short)((1L << (qbits - 1)) - 1); /* :5446 */
    const int ialpha = (int)(alpha * (1L << 4));               /* :5458 */
    int iter, j, e, n, v, parity = 0;
    short *soft = c->isoft, *iy = c->iy;

    memset(c->imin1, 0, sizeof(short) * R);
    memset(c->imin2, 0, sizeof(short) * R);
    memset(c->pos, 0, sizeof(int) * R);
    memset(c->sgn, 0, R);
    {                                              /* :5472-5500 energy-normalised quantiser */
        double en = 0, coef;
        for (v = 0; v < N; v++) en += y[v] * y[v];
        coef = sqrt(N / en);
        for (v = 0; v < N; v++) {
            double val = y[v];
            int sign = 0, ival;
            if (val < 0) { val = -val; sign = 1; }
            val *= coef;
            if (val > thr) val = thr;
            ival = (short)floor(val * max_quant / thr + 0.5);
            iy[v] = (short)(sign ? -ival : ival);
        }
    }
    memset(c->S, 0, (size_t)c->ne * M);
    if (maxsteps <= 0) return 1; /* parity is uninitialised in the reference here; treated as converged */

    for (iter = 0; iter < maxsteps; iter++) {
        memset(soft, 0, sizeof(short) * N);
        for (e = 0; e < c->ne; e++) {              /* STATE 1 :5540-5576, saturating after every add */
            const int jj = c->e_row[e], k = c->e_col[e], sh = c->e_shift[e];
            const short *m1 = c->imin1 + jj * M, *m2 = c->imin2 + jj * M;
            const int *ps = c->pos + jj * M;
            const unsigned char *sg = c->sgn + jj * M, *S = c->S + (size_t)e * M;
            short *col = soft + k * M;
            for (n = 0; n < M; n++) {
                short a = (ps[n] == k) ? m2[n] : m1[n];
                short m;
                int t = ROT(n, sh, M);
                a = (short)((a * ialpha) >> 4);
                m = (short)((S[n] ^ sg[n]) ? -a : a);
                col[t] = orc_limit((short)(col[t] + m), max_data);
            }
        }
        for (v = 0; v < N; v++) {                  /* STATE 2 :5579-5604 (no alpha here) */
            soft[v] = (short)(iy[v] + soft[v]);
            soft[v] = orc_limit(soft[v], max_data);
            decword[v] = decision ? (double)soft[v] : (double)(soft[v] < 0);
        }
        memset(c->synd, 0, R);
        for (j = 0; j < rh; j++) {                 /* STATE 3 :5610-5678 */
            short *m1 = c->imin1 + j * M, *m2 = c->imin2 + j * M;
            int *ps = c->pos + j * M;
            unsigned char *sg = c->sgn + j * M, *sy = c->synd + j * M;
            for (n = 0; n < M; n++) { c->inmin1[n] = max_data; c->inmin2[n] = max_data; c->npos[n] = 0; c->nsgn[n] = 0; }
            for (e = c->row_start[j]; e < c->row_start[j + 1]; e++) {
                const int k = c->e_col[e], sh = c->e_shift[e];
                const short *col = soft + k * M;
                unsigned char *S = c->S + (size_t)e * M;
                for (n = 0; n < M; n++) {
                    short r = col[ROT(n, sh, M)];
                    short a = (ps[n] == k) ? m2[n] : m1[n];
                    short val = (short)((a * ialpha) >> 4);
                    short t = (short)((S[n] ^ sg[n]) ? -val : val);
                    short msg = (short)(r - t);
                    unsigned char sn = (msg < 0);
                    sy[n] ^= (r < 0);
                    S[n] = sn;
                    c->nsgn[n] ^= sn;
                    val = (short)(msg < 0 ? -msg : msg);
                    val = (val > max_data) ? max_data : val;
                    if (val < c->inmin1[n]) { c->npos[n] = k; c->inmin2[n] = c->inmin1[n]; c->inmin1[n] = val; }
                    else if (val < c->inmin2[n]) c->inmin2[n] = val;
                }
            }
            for (n = 0; n < M; n++) { m1[n] = c->inmin1[n]; m2[n] = c->inmin2[n]; ps[n] = c->npos[n]; sg[n] = c->nsgn[n]; }
        }
        parity = 0;
        for (n = 0; n < R; n++) parity |= c->synd[n];
        if (!parity) break;
    }
    return parity ? -iter : iter + 1;             /* :5689 */
}

/* ---------------------------------------------------------------------------------------------
 * QAM mapper: QAM_modulator.cpp:69-194.  Per symbol of m bits: first m/2 bits -> I rail,
 * last m/2 -> Q rail, MSB first (p = {2^(m/2-1) .. 1}, :102-103); level = 2*gray[z] - s[m/2].
 * ------------------------------------------------------------------------------------------- */
static int orc_log2(int Q) { int m = 0; while ((1 << m) < Q) m++; return m; }

int orc_qam_modulate(int Q, const double *bits, int nbits, double *out) {
    static const short gray[16] = {0, 1, 3, 2, 7, 6, 4, 5, 15, 14, 12, 13, 8, 9, 11, 10}; /* :127 ("anti-gray") */
    static const short soff[5] = {0, 1, 3, 7, 15};                                        /* :128 */
    const int m = orc_log2(Q), h = m / 2;
    const int ns = (nbits + m - 1) / m;
    int sym, i;
    for (sym = 0; sym < ns; sym++) {
        int z1 = 0, z2 = 0;
        for (i = 0; i < h; i++) {
            int b1 = sym * m + i, b2 = sym * m + h + i;
            /* (int)(p * in): the reference pads the tail with zeros (bp_simulation.cpp:575) */
            z1 += (int)((1 << (h - 1 - i)) * (b1 < nbits ? bits[b1] : 0.0));
            z2 += (int)((1 << (h - 1 - i)) * (b2 < nbits ? bits[b2] : 0.0));
        }
        out[2 * sym] = (double)(2 * gray[z1] - soff[h]);
        out[2 * sym + 1] = (double)(2 * gray[z2] - soff[h]);
    }
    return ns;
}

/* ---------------------------------------------------------------------------------------------
 * Soft demapper: QAM_demodulator.cpp:99-566.  m == 2 (:113-139) and m == 4 (:203-275).
 * ------------------------------------------------------------------------------------------- */
static double orc_llr_or_p(double p0, double p1, double T, int out_type) {
    if (p0 == 0.0) return out_type == 0 ? T : 1.0;
    if (p1 == 0.0) return out_type == 0 ? -T : 0.0;
    return out_type == 0 ? log(p1 / p0) : p1;
}

void orc_qam_demodulate(int Q, double T, double sigma, const double *x, int ns, double *out, int out_type) {
    const int m = orc_log2(Q);
    int sym, rail, i;
    if (m == 2) {
        const double sigma2 = sigma * sigma;
        const int n = 2 * ns;
        for (sym = 0; sym < ns; sym++) {
            out[2 * sym] = 2.0 * x[2 * sym] / sigma2;
            out[2 * sym + 1] = 2.0 * x[2 * sym + 1] / sigma2;
        }
        if (out_type) {
            double P = 0.0;
            for (i = 0; i < n; i++) { out[i] = exp(out[i]); P += out[i]; }
            for (i = 0; i < n; i++) out[i] /= P;
        }
        return;
    }
    {
        const double N0 = 2.0 * sigma * sigma;   /* :142 */
        const int SQ = 1 << (m / 2);
        static const int soff[5] = {0, 1, 3, 7, 15};
        double P[16];
        for (sym = 0; sym < ns; sym++) {
            int hh = 0;
            for (rail = 0; rail < 2; rail++) {
                double sum = 0;
                for (i = 0; i < SQ; i++) {
                    double tmp = x[2 * sym + rail] - (double)(2 * i - soff[m / 2]);
                    tmp *= tmp;
                    tmp /= N0;
                    P[i] = (tmp < T) ? exp(-tmp) : 0.0;
                    sum += P[i];
                }
                for (i = 0; i < SQ; i++) P[i] /= sum;
                if (m == 4) {                     /* :203-275: levels ordered 00 01 11 10 */
                    out[sym * m + hh++] = orc_llr_or_p(P[0] + P[1], P[2] + P[3], T, out_type);
                    out[sym * m + hh++] = orc_llr_or_p(P[0] + P[3], P[1] + P[2], T, out_type);
                } else if (m == 6) {              /* :276-393 QAM-64, sums in the reference's association order */
                    const double p12 = P[0] + P[1], p34 = P[2] + P[3], p56 = P[4] + P[5], p78 = P[6] + P[7];
                    const double p1234 = p12 + p34, p5678 = p56 + p78, p1278 = p12 + p78, p3456 = p34 + p56;
                    out[sym * m + hh++] = orc_llr_or_p(p1234, p5678, T, out_type);
                    out[sym * m + hh++] = orc_llr_or_p(p1278, p3456, T, out_type);
                    out[sym * m + hh++] = orc_llr_or_p(P[0] + P[3] + P[4] + P[7], P[1] + P[2] + P[5] + P[6], T, out_type);
                } else if (m == 8) {              /* :395-561 QAM-256 */
                    const double p12 = P[0] + P[1], p34 = P[2] + P[3], p56 = P[4] + P[5], p78 = P[6] + P[7];
                    const double p9A = P[8] + P[9], pBC = P[10] + P[11], pDE = P[12] + P[13], pFG = P[14] + P[15];
                    const double p1234 = p12 + p34, p5678 = p56 + p78, p9ABC = p9A + pBC, pDEFG = pDE + pFG;
                    const double p1to8 = p1234 + p5678, p9toG = p9ABC + pDEFG;
                    out[sym * m + hh++] = orc_llr_or_p(p1to8, p9toG, T, out_type);
                    out[sym * m + hh++] = orc_llr_or_p(p1234 + pDEFG, p5678 + p9ABC, T, out_type);
                    out[sym * m + hh++] = orc_llr_or_p(p12 + p78 + p9A + pFG, p34 + p56 + pBC + pDE, T, out_type);
                    out[sym * m + hh++] = orc_llr_or_p(P[0] + P[3] + P[4] + P[7] + P[8] + P[11] + P[12] + P[15],
                                                       P[1] + P[2] + P[5] + P[6] + P[9] + P[10] + P[13] + P[14], T, out_type);
                }
            }
        }
    }
}
