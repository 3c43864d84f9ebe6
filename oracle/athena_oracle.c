/*
 * athena_oracle.c -- CPU restatement of athena's message-passing arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing here is shipped or measured as the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and there only as the checker / reported CPU
 * baseline.  The product path (athena_amd/csrc) never links or calls it.
 *
 * Every function restates, loop for loop and in the same fp32 accumulation
 * order, one procedure of the reference (nedtaylor/athena v2.1.1, paths below
 * relative to /root/reference/src/athena/).  Arrays keep the Fortran
 * conventions of the reference so the loops read the same:
 *   - val(F, N) column-major  ==  C row-major [N][F]  (vertex row contiguous)
 *   - adj_ia(N+1)   1-based row pointers
 *   - adj_ja(2,nnz) column-major, adj_ja(1,w)=neighbour (1-based),
 *                   adj_ja(2,w)=edge-feature column (1-based; 0 = none)
 * Arithmetic is strict IEEE fp32: compile with -ffp-contract=off so that
 * a*b+c is a rounded multiply followed by a rounded add, exactly what an
 * x86-64 baseline gfortran build of the reference executes.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   kipf_propagate fwd/bwd  -- pinned by the reference's own known-answer
 *       test test/test_diffstruc_extd_kipf.f90:22-45,73-91 (tests/golden).
 *   duvenaud_*, gno_*       -- the reference's tests hold no numeric vectors
 *       for these (shape/property checks only); pinned by the values the
 *       survey recorded from the reference where they exist,
 *       otherwise "parity unpinned" beyond the line-by-line restatement.
 *   matmul / softmax-sum / '+'  -- live in diffstruc v1.2.0 (fpm.toml:20),
 *       not in /root/reference: restated as plain fp32 matmul; unpinned.
 * Corroboration that is NOT a pin (a reference build over a stand-in for its
 * un-vendored dependencies pins nothing by the rules): the CPU suite runs
 * athena's own compiled kipf / duvenaud / graph_nop layers over such a stand-in
 * (scripts/integration_check/run_layers stock) and oracle/layers.py, which
 * composes these functions, agrees with their outputs and gradients at 1e-5
 * (tests/test_integration_compile.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define JA(k, w) adj_ja[2 * (size_t)(w) + (k)] /* adj_ja(k+1, w+1), values 1-based */

/* ------------------------------------------------------------------------- *
 * kipf_propagate            athena_diffstruc_extd_sub_kipf.f90:7-59
 *   c(:,v) = sum_w coeff * x(:, ja(1,w)),
 *   coeff = ((ia(v+1)-ia(v)) * (ia(u+1)-ia(u))) ** (-0.5_real32)   (:39-44)
 *   integer product first, then integer**real32 => powf(float(prod), -0.5f)
 * ------------------------------------------------------------------------- */
void oracle_kipf_propagate(int F, int N, const float *x, const int32_t *adj_ia,
                           const int32_t *adj_ja, float *c)
{
    for (int v = 0; v < N; ++v) {
        float *cv = c + (size_t)v * F;
        for (int i = 0; i < F; ++i) cv[i] = 0.0f;                 /* :31 */
        for (int w = adj_ia[v] - 1; w < adj_ia[v + 1] - 1; ++w) { /* :32 */
            int u = JA(0, w) - 1;
            int prod = (adj_ia[v + 1] - adj_ia[v]) * (adj_ia[u + 1] - adj_ia[u]);
            float coeff = powf((float)prod, -0.5f);              /* :39-42 */
            const float *xu = x + (size_t)u * F;
            for (int i = 0; i < F; ++i) cv[i] = cv[i] + coeff * xu[i]; /* :44 */
        }
    }
}

/* same op on a rectangular block (row partition with halo columns): degrees
 * are supplied instead of derived from adj_ia.  Not in the reference; used
 * only to check the multi-GPU shards. deg arrays are plain counts. */
void oracle_kipf_propagate_rect(int F, int n_rows, const float *x, const int32_t *adj_ia,
                                const int32_t *adj_ja, const int32_t *row_deg,
                                const int32_t *col_deg, float *c)
{
    for (int v = 0; v < n_rows; ++v) {
        float *cv = c + (size_t)v * F;
        for (int i = 0; i < F; ++i) cv[i] = 0.0f;
        for (int w = adj_ia[v] - 1; w < adj_ia[v + 1] - 1; ++w) {
            int u = JA(0, w) - 1;
            float coeff = powf((float)(row_deg[v] * col_deg[u]), -0.5f);
            const float *xu = x + (size_t)u * F;
            for (int i = 0; i < F; ++i) cv[i] = cv[i] + coeff * xu[i];
        }
    }
}

/* ------------------------------------------------------------------------- *
 * get_partial_kipf_propagate_left_val   ..._sub_kipf.f90:85-111
 *   output = 0; output(:, ja(1,w)) += upstream(:, v)      -- NO coeff (F5)
 * exact != 0 multiplies by coeff (the mathematically correct gradient; not
 * what the reference computes -- kept behind a flag).
 * n_out = number of columns of the output (== N for square graphs).
 * ------------------------------------------------------------------------- */
void oracle_kipf_propagate_bwd(int F, int N, int n_out, const float *g, const int32_t *adj_ia,
                               const int32_t *adj_ja, int exact, float *out)
{
    memset(out, 0, sizeof(float) * (size_t)n_out * F);            /* :100 */
    for (int v = 0; v < N; ++v) {                                 /* :101 */
        const float *gv = g + (size_t)v * F;
        for (int w = adj_ia[v] - 1; w < adj_ia[v + 1] - 1; ++w) { /* :102 */
            int u = JA(0, w) - 1;
            float *ou = out + (size_t)u * F;
            if (exact) {
                int prod = (adj_ia[v + 1] - adj_ia[v]) * (adj_ia[u + 1] - adj_ia[u]);
                float coeff = powf((float)prod, -0.5f);
                for (int i = 0; i < F; ++i) ou[i] = ou[i] + coeff * gv[i];
            } else {
                for (int i = 0; i < F; ++i) ou[i] = ou[i] + gv[i]; /* :104-105 */
            }
        }
    }
}

/* ------------------------------------------------------------------------- *
 * matmul(params(t), ptr2)    call site athena_kipf_msgpass_layer.f90:951
 * (diffstruc v1.2.0, not in tree -- plain fp32 matmul, k-ordered sum)
 *   Z(o,v) = sum_i W(o,i) P(i,v);  W(Fo,Fi) column-major flat = params%val(:,1)
 * reverse: dW(o,i) = sum_v dZ(o,v) P(i,v);  dP(i,v) = sum_o W(o,i) dZ(o,v)
 * ------------------------------------------------------------------------- */
void oracle_matmul(int Fo, int Fi, int N, const float *W, const float *P, float *Z)
{
    /* per output element the sum runs over i ascending from 0 (loops arranged o-innermost so
     * the compiler can vectorise across o without re-associating any sum) */
    for (int v = 0; v < N; ++v) {
        float *z = Z + (size_t)v * Fo;
        for (int o = 0; o < Fo; ++o) z[o] = 0.0f;
        for (int i = 0; i < Fi; ++i) {
            const float p = P[(size_t)v * Fi + i];
            const float *w = W + (size_t)Fo * i;
            for (int o = 0; o < Fo; ++o) z[o] = z[o] + w[o] * p;
        }
    }
}
void oracle_matmul_dw(int Fo, int Fi, int N, const float *dZ, const float *P, float *dW)
{
    /* per output element the sum runs over v ascending */
    for (size_t t = 0; t < (size_t)Fo * Fi; ++t) dW[t] = 0.0f;
    for (int v = 0; v < N; ++v)
        for (int i = 0; i < Fi; ++i) {
            const float p = P[(size_t)v * Fi + i];
            const float *dz = dZ + (size_t)v * Fo;
            float *w = dW + (size_t)Fo * i;
            for (int o = 0; o < Fo; ++o) w[o] = w[o] + dz[o] * p;
        }
}
void oracle_matmul_dx(int Fo, int Fi, int N, const float *W, const float *dZ, float *dP)
{
    for (int v = 0; v < N; ++v)
        for (int i = 0; i < Fi; ++i) {
            float s = 0.0f;
            for (int o = 0; o < Fo; ++o) s = s + W[o + (size_t)Fo * i] * dZ[(size_t)v * Fo + o];
            dP[(size_t)v * Fi + i] = s;
        }
}

/* ------------------------------------------------------------------------- *
 * duvenaud_propagate        athena_diffstruc_extd_sub_duvenaud.f90:7-59
 *   c(:,v) = sum_w [ x(:, ja(1,w)) ; e(:, ja(2,w)) ]            (:34-42)
 * ja(2,w) == 0 would index e(:,0) (out of bounds in the reference);
 * here such an entry contributes a zero edge vector (documented deviation).
 * ------------------------------------------------------------------------- */
void oracle_duvenaud_propagate(int Fv, int Fe, int N, const float *x, const float *e,
                               const int32_t *adj_ia, const int32_t *adj_ja, float *c)
{
    int Fc = Fv + Fe;
    for (int v = 0; v < N; ++v) {
        float *cv = c + (size_t)v * Fc;
        for (int i = 0; i < Fc; ++i) cv[i] = 0.0f;                /* :35 */
        for (int w = adj_ia[v] - 1; w < adj_ia[v + 1] - 1; ++w) { /* :36 */
            int u = JA(0, w) - 1, ed = JA(1, w) - 1;
            const float *xu = x + (size_t)u * Fv;
            for (int i = 0; i < Fv; ++i) cv[i] = cv[i] + xu[i];
            if (ed >= 0) {
                const float *ee = e + (size_t)ed * Fe;
                for (int i = 0; i < Fe; ++i) cv[Fv + i] = cv[Fv + i] + ee[i];
            }
        }
    }
}
/* get_partial_duvenaud_propagate_left_val   :115-141  dx(:,ja(1,w)) += g(1:Fv,v) */
void oracle_duvenaud_propagate_bwd_x(int Fv, int Fe, int N, const float *g, const int32_t *adj_ia,
                                     const int32_t *adj_ja, float *dx)
{
    int Fc = Fv + Fe;
    memset(dx, 0, sizeof(float) * (size_t)N * Fv);
    for (int v = 0; v < N; ++v)
        for (int w = adj_ia[v] - 1; w < adj_ia[v + 1] - 1; ++w) {
            int u = JA(0, w) - 1;
            for (int i = 0; i < Fv; ++i) dx[(size_t)u * Fv + i] += g[(size_t)v * Fc + i];
        }
}
/* get_partial_duvenaud_propagate_right_val  :143-171  de(:,ja(2,w)) += g(Fv+1:,v) */
void oracle_duvenaud_propagate_bwd_e(int Fv, int Fe, int N, int E, const float *g,
                                     const int32_t *adj_ia, const int32_t *adj_ja, float *de)
{
    int Fc = Fv + Fe;
    memset(de, 0, sizeof(float) * (size_t)E * Fe);
    for (int v = 0; v < N; ++v)
        for (int w = adj_ia[v] - 1; w < adj_ia[v + 1] - 1; ++w) {
            int ed = JA(1, w) - 1;
            if (ed < 0) continue;
            for (int i = 0; i < Fe; ++i) de[(size_t)ed * Fe + i] += g[(size_t)v * Fc + Fv + i];
        }
}

/* ------------------------------------------------------------------------- *
 * duvenaud_update           ..._sub_duvenaud.f90:176-228
 *   d = max(min_deg, min(deg_v, max_deg)) - min_deg + 1          (:205-206)
 *   W_d(Fo,Fi) = weight(interval*(d-1)+1 : interval*d)           (:207-208)
 *   c(:,v) = matmul(W_d, a(:,v) / real(d))   -- bucket index is the divisor
 * ------------------------------------------------------------------------- */
static inline int duv_bucket(const int32_t *adj_ia, int v, int min_deg, int max_deg)
{
    int deg = adj_ia[v + 1] - adj_ia[v];
    int m = deg < max_deg ? deg : max_deg;
    int d = (min_deg > m ? min_deg : m) - min_deg + 1;
    return d;
}
void oracle_duvenaud_update(int Fo, int Fi, int N, const float *a, const float *weight,
                            const int32_t *adj_ia, int min_deg, int max_deg, float *c)
{
    size_t interval = (size_t)Fo * Fi;
    float *tmp = (float *)malloc(sizeof(float) * Fi);
    for (int v = 0; v < N; ++v) {
        int d = duv_bucket(adj_ia, v, min_deg, max_deg);
        const float *Wd = weight + interval * (d - 1);
        float rd = (float)d;
        for (int i = 0; i < Fi; ++i) tmp[i] = a[(size_t)v * Fi + i] / rd;
        for (int o = 0; o < Fo; ++o) {
            float s = 0.0f;
            for (int i = 0; i < Fi; ++i) s = s + Wd[o + (size_t)Fo * i] * tmp[i];
            c[(size_t)v * Fo + o] = s;
        }
    }
    free(tmp);
}
/* get_partial_duvenaud_update_val  :284-324
 *   da(:,v) = matmul(g(:,v), W_d) / real(d) */
void oracle_duvenaud_update_bwd_a(int Fo, int Fi, int N, const float *g, const float *weight,
                                  const int32_t *adj_ia, int min_deg, int max_deg, float *da)
{
    size_t interval = (size_t)Fo * Fi;
    for (int v = 0; v < N; ++v) {
        int d = duv_bucket(adj_ia, v, min_deg, max_deg);
        const float *Wd = weight + interval * (d - 1);
        float rd = (float)d;
        for (int i = 0; i < Fi; ++i) {
            float s = 0.0f;
            for (int o = 0; o < Fo; ++o) s = s + g[(size_t)v * Fo + o] * Wd[o + (size_t)Fo * i];
            da[(size_t)v * Fi + i] = s / rd;
        }
    }
}
/* get_partial_duvenaud_update_weight_val  :326-368
 *   dW(d_off + i + Fo*(j-1)) += g(i,v) * a(j,v) / real(d)        (:361-365)
 * n_buckets = max_deg - min_deg + 1 (size of the packed weight / Fo / Fi) */
void oracle_duvenaud_update_bwd_w(int Fo, int Fi, int N, const float *g, const float *a,
                                  const int32_t *adj_ia, int min_deg, int max_deg, float *dW)
{
    size_t interval = (size_t)Fo * Fi;
    int nb = max_deg - min_deg + 1;
    memset(dW, 0, sizeof(float) * interval * nb);
    for (int v = 0; v < N; ++v) {
        int d = duv_bucket(adj_ia, v, min_deg, max_deg);
        float rd = (float)d;
        float *o_ = dW + interval * (d - 1);
        for (int j = 0; j < Fi; ++j)
            for (int i = 0; i < Fo; ++i)
                o_[i + (size_t)Fo * j] =
                    o_[i + (size_t)Fo * j] + g[(size_t)v * Fo + i] * a[(size_t)v * Fi + j] / rd;
    }
}

/* ------------------------------------------------------------------------- *
 * element-wise activations used by the three layers' defaults
 *   none    athena_activation_none.f90        sigmoid athena_activation_sigmoid.f90
 *   relu    athena_activation_relu.f90:186-205  (max(val, 0))
 *   softmax athena_diffstruc_extd_sub.f90:295-331 (dim=2: per column/vertex)
 * kind: 0 none, 1 relu, 2 sigmoid, 3 tanh
 * ------------------------------------------------------------------------- */
void oracle_activation(int kind, size_t n, const float *z, float *y)
{
    for (size_t k = 0; k < n; ++k) {
        float t = z[k];
        switch (kind) {
        case 1: y[k] = t > 0.0f ? t : 0.0f; break;
        case 2: y[k] = 1.0f / (1.0f + expf(-t)); break;
        case 3: y[k] = tanhf(t); break;
        default: y[k] = t;
        }
    }
}
/* gradient wrt pre-activation given the activation OUTPUT y and upstream g */
void oracle_activation_bwd(int kind, size_t n, const float *y, const float *g, float *dz)
{
    for (size_t k = 0; k < n; ++k) {
        switch (kind) {
        case 1: dz[k] = y[k] > 0.0f ? g[k] : 0.0f; break;
        case 2: dz[k] = g[k] * y[k] * (1.0f - y[k]); break;
        case 3: dz[k] = g[k] * (1.0f - y[k] * y[k]); break;
        default: dz[k] = g[k];
        }
    }
}
/* softmax_array dim=2 (:309-313): per column (vertex) over the F rows */
void oracle_softmax_cols(int F, int N, const float *z, float *y)
{
    for (int v = 0; v < N; ++v) {
        const float *zv = z + (size_t)v * F;
        float *yv = y + (size_t)v * F;
        float m = zv[0];
        for (int i = 1; i < F; ++i) m = zv[i] > m ? zv[i] : m;
        float s = 0.0f;
        for (int i = 0; i < F; ++i) { yv[i] = expf(zv[i] - m); s = s + yv[i]; }
        for (int i = 0; i < F; ++i) yv[i] = yv[i] / s;
    }
}
/* softmax backward per column: dz = y * (g - sum(g*y)) */
void oracle_softmax_cols_bwd(int F, int N, const float *y, const float *g, float *dz)
{
    for (int v = 0; v < N; ++v) {
        const float *yv = y + (size_t)v * F, *gv = g + (size_t)v * F;
        /* get_partial_softmax_val, athena_diffstruc_extd_sub.f90:358-379: output = val*grad;
         * output(:,s) = output(:,s) - val(:,s)*sum(output(:,s)) */
        float dot = 0.0f;
        for (int i = 0; i < F; ++i) dot = dot + yv[i] * gv[i];
        for (int i = 0; i < F; ++i) dz[(size_t)v * F + i] = yv[i] * gv[i] - yv[i] * dot;
    }
}

/* ------------------------------------------------------------------------- *
 * Duvenaud readout    athena_duvenaud_msgpass_layer.f90:838-855
 *   out(:,s) += sum_{v in graph s} act_readout( R_t z_t(:,v) )
 * Here for ONE time step over a block-diagonal batch: seg(S+1) 0-based vertex
 * offsets per graph; p(O,N) already-activated per-vertex values.
 * accumulate != 0 adds into out (the `ptr3 + sum(...)` of :849-852).
 * ------------------------------------------------------------------------- */
void oracle_segment_sum(int O, int S, const int32_t *seg, const float *p, int accumulate, float *out)
{
    for (int s = 0; s < S; ++s) {
        float *os = out + (size_t)s * O;
        for (int o = 0; o < O; ++o) {
            float acc = 0.0f;
            for (int v = seg[s]; v < seg[s + 1]; ++v) acc = acc + p[(size_t)v * O + o];
            os[o] = accumulate ? os[o] + acc : acc;
        }
    }
}

/* ------------------------------------------------------------------------- *
 * gno_kernel_eval     athena_diffstruc_extd_sub_nop.f90:26-115
 *   theta = [ U(H,d) | b_u(H) | V(F,H) | b_v(F) ], F = F_out*F_in  (:74-82)
 *   hidden = max(U dx + b_u, 0);  kappa(:,e) = V hidden + b_v      (:88-93)
 * ------------------------------------------------------------------------- */
void oracle_gno_kernel_eval(int d, int H, int F, int E, const float *coords, const float *theta,
                            float *kappa)
{
    const float *U = theta, *bu = theta + (size_t)H * d, *V = bu + H, *bv = V + (size_t)F * H;
    float *hid = (float *)malloc(sizeof(float) * H);
    for (int e = 0; e < E; ++e) {
        const float *dx = coords + (size_t)e * d;
        for (int k = 0; k < H; ++k) {
            float s = 0.0f;
            for (int j = 0; j < d; ++j) s = s + U[k + (size_t)H * j] * dx[j];
            s = s + bu[k];
            hid[k] = s > 0.0f ? s : 0.0f;
        }
        float *ke = kappa + (size_t)e * F;
        for (int f = 0; f < F; ++f) {
            float s = 0.0f;
            for (int k = 0; k < H; ++k) s = s + V[f + (size_t)F * k] * hid[k];
            ke[f] = s + bv[f];
        }
    }
    free(hid);
}

/* gno_aggregate       ..._sub_nop.f90:330-397
 *   c(:,i) += reshape(kappa(:,ja(2,jj)),[Fo,Fi]) @ x(:,ja(1,jj))   (:367-378) */
void oracle_gno_aggregate(int Fi, int Fo, int N, const float *x, const float *kappa,
                          const int32_t *adj_ia, const int32_t *adj_ja, float *c)
{
    memset(c, 0, sizeof(float) * (size_t)N * Fo);
    for (int i = 0; i < N; ++i)
        for (int jj = adj_ia[i] - 1; jj < adj_ia[i + 1] - 1; ++jj) {
            int j = JA(0, jj) - 1, e = JA(1, jj) - 1;
            if (e < 0) continue; /* edge id 0 (self-loop entry): the reference would index kernel column 0, out of
                                  * bounds; defined here, as for duvenaud_propagate, as a zero kernel */
            const float *K = kappa + (size_t)e * Fo * Fi;
            const float *xj = x + (size_t)j * Fi;
            for (int o = 0; o < Fo; ++o) {
                float s = 0.0f;
                for (int q = 0; q < Fi; ++q) s = s + K[o + (size_t)Fo * q] * xj[q];
                c[(size_t)i * Fo + o] = c[(size_t)i * Fo + o] + s;
            }
        }
}
/* get_partial_gno_agg_features_val  :419-458   dx(:,j) += K_e^T g(:,i) */
void oracle_gno_aggregate_bwd_x(int Fi, int Fo, int N, const float *g, const float *kappa,
                                const int32_t *adj_ia, const int32_t *adj_ja, float *dx)
{
    memset(dx, 0, sizeof(float) * (size_t)N * Fi);
    for (int i = 0; i < N; ++i)
        for (int jj = adj_ia[i] - 1; jj < adj_ia[i + 1] - 1; ++jj) {
            int j = JA(0, jj) - 1, e = JA(1, jj) - 1;
            if (e < 0) continue; /* edge id 0 (self-loop entry): the reference would index kernel column 0, out of
                                  * bounds; defined here, as for duvenaud_propagate, as a zero kernel */
            const float *K = kappa + (size_t)e * Fo * Fi;
            for (int q = 0; q < Fi; ++q) {
                float s = 0.0f;
                for (int o = 0; o < Fo; ++o) s = s + K[o + (size_t)Fo * q] * g[(size_t)i * Fo + o];
                dx[(size_t)j * Fi + q] = dx[(size_t)j * Fi + q] + s;
            }
        }
}
/* get_partial_gno_agg_kernels_val   :480-526   dkappa((fi-1)*Fo+fo, e) += g(fo,i) x(fi,j) */
void oracle_gno_aggregate_bwd_k(int Fi, int Fo, int N, int E, const float *g, const float *x,
                                const int32_t *adj_ia, const int32_t *adj_ja, float *dkappa)
{
    memset(dkappa, 0, sizeof(float) * (size_t)E * Fo * Fi);
    for (int i = 0; i < N; ++i)
        for (int jj = adj_ia[i] - 1; jj < adj_ia[i + 1] - 1; ++jj) {
            int j = JA(0, jj) - 1, e = JA(1, jj) - 1;
            if (e < 0) continue; /* edge id 0 (self-loop entry): the reference would index kernel column 0, out of
                                  * bounds; defined here, as for duvenaud_propagate, as a zero kernel */
            float *dk = dkappa + (size_t)e * Fo * Fi;
            for (int q = 0; q < Fi; ++q)
                for (int o = 0; o < Fo; ++o)
                    dk[(size_t)q * Fo + o] =
                        dk[(size_t)q * Fo + o] + g[(size_t)i * Fo + o] * x[(size_t)j * Fi + q];
        }
}
/* get_partial_gno_kernel_params_val :235-325  (dtheta from dkappa) */
void oracle_gno_kernel_bwd_theta(int d, int H, int F, int E, const float *coords,
                                 const float *theta, const float *dkappa, float *dtheta)
{
    const float *U = theta, *bu = theta + (size_t)H * d, *V = bu + H;
    size_t off_bu = (size_t)H * d, off_V = off_bu + H, off_bv = off_V + (size_t)F * H;
    memset(dtheta, 0, sizeof(float) * (off_bv + F));
    float *pre = (float *)malloc(sizeof(float) * H), *hid = (float *)malloc(sizeof(float) * H),
          *gh = (float *)malloc(sizeof(float) * H);
    for (int e = 0; e < E; ++e) {
        const float *dx = coords + (size_t)e * d, *up = dkappa + (size_t)e * F;
        for (int k = 0; k < H; ++k) {
            float s = 0.0f;
            for (int j = 0; j < d; ++j) s = s + U[k + (size_t)H * j] * dx[j];
            pre[k] = s + bu[k];
            hid[k] = pre[k] > 0.0f ? pre[k] : 0.0f;
        }
        for (int f = 0; f < F; ++f) dtheta[off_bv + f] = dtheta[off_bv + f] + up[f];   /* :291 */
        for (int k = 0; k < H; ++k)                                                    /* :294-300 */
            for (int f = 0; f < F; ++f)
                dtheta[off_V + (size_t)k * F + f] = dtheta[off_V + (size_t)k * F + f] + up[f] * hid[k];
        for (int k = 0; k < H; ++k) {                                                  /* :303-306 */
            float s = 0.0f;
            for (int f = 0; f < F; ++f) s = s + V[f + (size_t)F * k] * up[f];
            gh[k] = pre[k] <= 0.0f ? 0.0f : s;
        }
        for (int k = 0; k < H; ++k) dtheta[off_bu + k] = dtheta[off_bu + k] + gh[k];   /* :309 */
        for (int j = 0; j < d; ++j)                                                    /* :312-318 */
            for (int k = 0; k < H; ++k)
                dtheta[(size_t)j * H + k] = dtheta[(size_t)j * H + k] + gh[k] * dx[j];
    }
    free(pre); free(hid); free(gh);
}
/* get_partial_gno_kernel_coords_val :137-216  dcoords(:,e) = (V diag(mask) U)^T dkappa(:,e) */
void oracle_gno_kernel_bwd_coords(int d, int H, int F, int E, const float *coords,
                                  const float *theta, const float *dkappa, float *dcoords)
{
    const float *U = theta, *bu = theta + (size_t)H * d, *V = bu + H;
    float *J = (float *)malloc(sizeof(float) * (size_t)F * d);
    for (int e = 0; e < E; ++e) {
        const float *dx = coords + (size_t)e * d, *up = dkappa + (size_t)e * F;
        memset(J, 0, sizeof(float) * (size_t)F * d);
        for (int k = 0; k < H; ++k) {
            float s = 0.0f;
            for (int j = 0; j < d; ++j) s = s + U[k + (size_t)H * j] * dx[j];
            s = s + bu[k];
            if (s > 0.0f)                                          /* :195, :202-206 */
                for (int j = 0; j < d; ++j)
                    for (int f = 0; f < F; ++f)
                        J[f + (size_t)F * j] = J[f + (size_t)F * j] + V[f + (size_t)F * k] * U[k + (size_t)H * j];
        }
        for (int j = 0; j < d; ++j) {                              /* :209-210 */
            float s = 0.0f;
            for (int f = 0; f < F; ++f) s = s + up[f] * J[f + (size_t)F * j];
            dcoords[(size_t)e * d + j] = s;
        }
    }
    free(J);
}

/* add_bias(dim=1): y(o,v) += b(o)   call site athena_graph_nop_layer.f90:768-772 */
void oracle_add_bias_rows(int F, int N, const float *b, float *y)
{
    for (int v = 0; v < N; ++v)
        for (int o = 0; o < F; ++o) y[(size_t)v * F + o] = y[(size_t)v * F + o] + b[o];
}

/* ========================================================================= *
 * Parameter update of a train step: the flat-vector path of
 * network_type%update, athena_network_sub.f90:2816-2929 -- gradients are
 * clipped (clip_dict%apply, :2903), then optimiser%minimise(params, gradients).
 * ========================================================================= */

/* apply_clip        athena_clipper.f90:165-210  (no bias argument: bias_ = [0])
 *   min/max clamp, then scale = min(1, norm / sqrt(sum(g**2) + sum(bias_)**2)) */
void oracle_clip(size_t n, float *g, int l_min_max, float mn, float mx, int l_norm, float norm)
{
    if (l_min_max)
        for (size_t i = 0; i < n; ++i) g[i] = fmaxf(mn, fminf(mx, g[i]));
    if (l_norm) {
        float s = 0.0f;
        for (size_t i = 0; i < n; ++i) s = s + g[i] * g[i];
        float scale = fminf(1.0f, norm / sqrtf(s + 0.0f));
        if (scale < 1.0f)
            for (size_t i = 0; i < n; ++i) g[i] = g[i] * scale;
    }
}

/* regularise_l1 / _l2 / _l1l2   athena_regulariser.f90:85-138
 *   kind 0 none, 1 l1, 2 l2, 3 l1l2;  sign(1,p) = +1 for p >= +0, -1 otherwise */
static inline float reg_term(int kind, float l1, float l2, float p, float lr)
{
    const float sg = signbit(p) ? -1.0f : 1.0f;
    switch (kind) {
    case 1: return lr * l1 * sg;
    case 2: return lr * 2.0f * l2 * p;
    case 3: return lr * (l1 * sg + 2.0f * l2 * p);
    default: return 0.0f;
    }
}

/* minimise_sgd      athena_optimiser.f90:634-673 */
void oracle_sgd_step(size_t n, float lr, float momentum, int nesterov, int reg_kind, float l1, float l2,
                     float *param, float *grad, float *velocity)
{
    for (size_t i = 0; i < n; ++i) {
        float g = grad[i];
        if (reg_kind) g = g + reg_term(reg_kind, l1, l2, param[i], lr);
        g = -lr * g;
        if (momentum > 1.0e-8f) {
            velocity[i] = momentum * velocity[i] + g;
            if (nesterov) param[i] = param[i] + momentum * velocity[i] + g;
            else param[i] = param[i] + velocity[i];
        } else {
            velocity[i] = g;
            param[i] = param[i] + velocity[i];
        }
        grad[i] = g;
    }
}

/* beta**iter with an INTEGER exponent (athena_optimiser.f90:1059-1060): compilers lower real**integer to
 * repeated multiplication (binary powering), not powf */
float oracle_powi(float x, int n)
{
    float r = (n & 1) ? x : 1.0f;
    for (n >>= 1; n; n >>= 1) {
        x = x * x;
        if (n & 1) r = r * x;
    }
    return r;
}

/* minimise_adam     athena_optimiser.f90:1027-1091
 *   reg_kind as above; decoupled: l2_regulariser_type%decoupled (AdamW branch :1069-1072);
 *   the classical-L2 branch (:1073-1079) adds l2*param to m_hat on top of the regularised gradient. */
void oracle_adam_step(size_t n, float lr, float beta1, float beta2, float epsilon, int iter, int reg_kind,
                      float l1, float l2, int decoupled, float *param, float *grad, float *m, float *v)
{
    const float bc1 = 1.0f - oracle_powi(beta1, iter), bc2 = 1.0f - oracle_powi(beta2, iter);
    for (size_t i = 0; i < n; ++i) {
        float g = grad[i];
        if (reg_kind) g = g + reg_term(reg_kind, l1, l2, param[i], lr);
        grad[i] = g;
        m[i] = beta1 * m[i] + (1.0f - beta1) * g;
        v[i] = beta2 * v[i] + (1.0f - beta2) * g * g;
        const float m_hat = m[i] / bc1, v_hat = v[i] / bc2;
        if (reg_kind == 2 && decoupled) {
            param[i] = param[i] - lr * l2 * param[i];
            param[i] = param[i] - lr * (m_hat / (sqrtf(v_hat) + epsilon));
        } else if (reg_kind == 2) {
            param[i] = param[i] - lr * ((m_hat + l2 * param[i]) / (sqrtf(v_hat) + epsilon));
        } else {
            param[i] = param[i] - lr * (m_hat / (sqrtf(v_hat) + epsilon));
        }
    }
}

/* compute_mse       athena_loss.f90:393-430:  mean(squared(predicted - expected)) / 2 over one array
 *   (diffstruc's mean/squared are outside /root/reference: taken as the mean over all elements; unpinned)
 *   returns the loss; dpred = d loss / d predicted = (p - e) / n */
float oracle_mse(size_t n, const float *pred, const float *expected, float *dpred)
{
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float d = pred[i] - expected[i];
        s = s + d * d;
        if (dpred) dpred[i] = d / (float)n;
    }
    return s / (float)n / 2.0f;
}


/* swish_array / get_partial_swish_val    athena_diffstruc_extd_sub.f90:424-455, :477-492
 *   y = x * (1 / (1 + exp(-beta*x)));  dx = g * e * (beta*x + e + 1) / (e + 1)**2, e = exp(beta*x)
 *   (the reverse pass differentiates at the INPUT x, and overflows to inf/inf = NaN exactly where
 *   the reference does, beta*x > ~88) */
void oracle_swish(size_t n, float beta, const float *x, float *y)
{
    for (size_t i = 0; i < n; ++i) y[i] = x[i] * (1.0f / (1.0f + expf(-beta * x[i])));
}
void oracle_swish_bwd(size_t n, float beta, const float *x, const float *g, float *dx)
{
    for (size_t i = 0; i < n; ++i) {
        const float e = expf(beta * x[i]);
        dx[i] = g[i] * e * (beta * x[i] + e + 1.0f) / ((e + 1.0f) * (e + 1.0f));
    }
}

/* activations with attributes: athena_activation_{linear,relu,sigmoid,tanh,leaky_relu,selu,gaussian,piecewise}.f90
 * `apply` -- y = f(x) * scale.  kind: 0 linear (:194), 1 relu max(val, threshold) (:201), 2 sigmoid, 3 tanh,
 * 4 leaky_relu max(val*alpha, val) (:204), 5 selu merge(val*lambda, (exp(val)-1)*alpha*lambda, val>0) (:227-229),
 * 6 gaussian (formula in the module header, athena_activation_gaussian.f90:8-11; the function itself lives in
 * diffstruc, absent here), 7 piecewise (piecewise_array, athena_diffstruc_extd_sub.f90:216-254).
 * Reverse factors, evaluated at the input: the elementary ones are diffstruc's (absent; the mathematical
 * derivative is restated, ties of max() unpinned); piecewise follows get_partial_piecewise_val :275-290 literally,
 * including its always-true test for limit >= 0. */
static float oracle_actp_f(int kind, float x, float p0, float p1)
{
    switch (kind) {
    case 1: return x > p0 ? x : p0;
    case 2: return 1.0f / (1.0f + expf(-x));
    case 3: return tanhf(x);
    case 4: return x * p0 > x ? x * p0 : x;
    case 5: return x > 0.0f ? x * p1 : (expf(x) - 1.0f) * p0 * p1;
    case 6: {
        const float t = (x - p1) / p0;
        return 1.0f / (sqrtf(6.283185307179586f) * p0) * expf(-0.5f * t * t);
    }
    case 7:
        if (x >= p1) return p0 * (x - p1) + p1;
        if (x <= -p1) return p0 * (x + p1) - p1;
        return x;
    default: return x;
    }
}
void oracle_activation_param(int kind, size_t n, float scale, float p0, float p1, const float *x, float *y)
{
    for (size_t i = 0; i < n; ++i) y[i] = oracle_actp_f(kind, x[i], p0, p1) * scale;
}
void oracle_activation_param_bwd(int kind, size_t n, float scale, float p0, float p1, const float *x, const float *g,
                                 float *dx)
{
    for (size_t i = 0; i < n; ++i) {
        const float u = g[i] * scale, v = x[i];
        float d;
        switch (kind) {
        case 1: d = v > p0 ? u : 0.0f; break;
        case 2: { const float y = 1.0f / (1.0f + expf(-v)); d = u * y * (1.0f - y); break; }
        case 3: { const float y = tanhf(v); d = u * (1.0f - y * y); break; }
        case 4: d = v * p0 > v ? u * p0 : u; break;
        case 5: d = v > 0.0f ? u * p1 : u * (expf(v) * p0 * p1); break;
        case 6: d = u * (-(v - p1) / (p0 * p0) * oracle_actp_f(6, v, p0, p1)); break;
        case 7: d = (v <= p1 || v >= -p1) ? u : u * p0; break;
        default: d = u;
        }
        dx[i] = d;
    }
}

/* 'concatenate' merge of two inputs of a layer (network%add(..., operator='concatenate'),
 * example/msgpass_euler/src/main.f90:192-255): features stacked per vertex, a first */
void oracle_concat(int N, int Fa, int Fb, const float *a, const float *b, float *out)
{
    for (int v = 0; v < N; ++v) {
        memcpy(out + (size_t)v * (Fa + Fb), a + (size_t)v * Fa, sizeof(float) * Fa);
        memcpy(out + (size_t)v * (Fa + Fb) + Fa, b + (size_t)v * Fb, sizeof(float) * Fb);
    }
}
