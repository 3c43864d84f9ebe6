/*
 * athena_oracle_omp.c -- the same Kipf layer step as athena_oracle.c, threaded over rows with OpenMP.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as athena_oracle.c).  The reference has no threading
 * (SURVEY.md F1); this file exists only so that bench.py can report, for context, what ALL host cores reach on
 * the same workload next to the faithful single-thread baseline ("(ii) OpenMP over rows on all
 * host cores for context").  Differences from the reference's loops, forced by threading:
 *   - the backward scatter (athena_diffstruc_extd_sub_kipf.f90:100-109) is evaluated in PULL form over a
 *     transposed CSR that the caller builds (sources ascending: the same per-element summation order);
 *   - dW is accumulated per thread and the partials are added in thread order (different association).
 * Forward aggregation and the two row-wise contractions are the reference's loops, rows shared out statically.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_omp_threads(void) { return omp_get_max_threads(); }

/* one interior Kipf layer step: P = A^ X; Z = P Wt; dW = dZ^T P; dP = dZ W; dX = scatter(dP) (no coefficient).
 * CSR 0-based (rowptr, col) with per-entry coefficient; transposed CSR (t_rowptr, t_src) for the pull. */
void oracle_omp_kipf_step(int N, int F, const int32_t *rowptr, const int32_t *col, const float *coef,
                          const int32_t *t_rowptr, const int32_t *t_src, const float *x, const float *Wt /* [F][F] */,
                          const float *dz, float *P, float *Z, float *dW, float *dP, float *dX)
{
    const int nt = omp_get_max_threads();
    float *part = (float *)calloc((size_t)nt * F * F, sizeof(float));
#pragma omp parallel
    {
        const int tid = omp_get_thread_num();
        float *mine = part + (size_t)tid * F * F;
#pragma omp for schedule(static)
        for (int v = 0; v < N; ++v) {
            float *p = P + (size_t)v * F;
            for (int i = 0; i < F; ++i) p[i] = 0.0f;
            for (int w = rowptr[v]; w < rowptr[v + 1]; ++w) {
                const float c = coef[w];
                const float *xu = x + (size_t)col[w] * F;
                for (int i = 0; i < F; ++i) p[i] = p[i] + c * xu[i];
            }
            float *z = Z + (size_t)v * F;
            const float *g = dz + (size_t)v * F;
            float *dp = dP + (size_t)v * F;
            for (int o = 0; o < F; ++o) z[o] = 0.0f;
            for (int i = 0; i < F; ++i) {
                const float pi = p[i];
                const float *wrow = Wt + (size_t)i * F;
                float s = 0.0f;
                for (int o = 0; o < F; ++o) {
                    z[o] = z[o] + pi * wrow[o];          /* Z = P Wt */
                    s = s + g[o] * wrow[o];              /* dP[v,i] = sum_o dZ[v,o] Wt[i,o] */
                    mine[(size_t)i * F + o] += pi * g[o]; /* dWt[i,o] += P[v,i] dZ[v,o] */
                }
                dp[i] = s;
            }
        }
#pragma omp for schedule(static)
        for (int u = 0; u < N; ++u) {
            float *d = dX + (size_t)u * F;
            for (int i = 0; i < F; ++i) d[i] = 0.0f;
            for (int w = t_rowptr[u]; w < t_rowptr[u + 1]; ++w) {
                const float *gp = dP + (size_t)t_src[w] * F;
                for (int i = 0; i < F; ++i) d[i] = d[i] + gp[i];
            }
        }
    }
    for (int i = 0; i < F * F; ++i) {
        float s = 0.0f;
        for (int t = 0; t < nt; ++t) s = s + part[(size_t)t * F * F + i];
        dW[i] = s;
    }
    free(part);
}
