// Host-pointer variants of the op-level entry points ("phase 1" of SURVEY.md 7.4): a Fortran caller
// that only holds array_type%val passes host arrays; the shim stages them through HBM around the same
// device entry point.  Plumbing, never timed -- device-resident callers use the *_dev entry points.
#include <initializer_list>

#include "common.h"

using namespace amp;

namespace {

struct Stage {
    const void *in;   // host source (nullptr: output only)
    void *out;        // host destination (nullptr: input only)
    size_t bytes;
    void *dev = nullptr;
};

// Staging buffers are pooled per argument position (grown on demand, kept until athena_mp_finalize): a
// hipMalloc / hipFree pair per argument costs more than the whole kernel for mini-batch sized inputs.
constexpr int kMaxStaged = 12;
void *g_pool[kMaxStaged] = {};
size_t g_pool_bytes[kMaxStaged] = {};

int pool_get(int slot, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 4;
    if (g_pool_bytes[slot] < bytes) {
        if (g_pool[slot]) {
            AMP_HIP(hipStreamSynchronize(stream()));
            AMP_HIP(hipFree(g_pool[slot]));
            g_pool[slot] = nullptr;
            g_pool_bytes[slot] = 0;
        }
        const size_t want = bytes + bytes / 4;   // headroom so a slowly growing batch does not reallocate every call
        AMP_HIP(hipMalloc(&g_pool[slot], want));
        g_pool_bytes[slot] = want;
    }
    *out = g_pool[slot];
    return 0;
}

// takes a pooled device buffer per argument, uploads inputs, runs body(dev pointers), downloads outputs
template <typename Body> int staged(std::initializer_list<Stage> args, Body &&body)
{
    std::vector<Stage> a(args);
    if ((int)a.size() > kMaxStaged) {
        set_error("host staging: too many arguments");
        return 1;
    }
    int rc = 0, slot = 0;
    for (auto &s : a) {
        if (pool_get(slot++, s.bytes, &s.dev)) {
            if (rc == 0) set_error("host staging: device allocation of %zu bytes failed", s.bytes);
            rc = 1;
            break;
        }
        if (s.in && s.bytes && hipMemcpyAsync(s.dev, s.in, s.bytes, hipMemcpyHostToDevice, stream()) != hipSuccess) {
            set_error("host staging: upload failed");
            rc = 1;
            break;
        }
    }
    if (rc == 0) {
        std::vector<void *> d;
        for (auto &s : a) d.push_back(s.dev);
        rc = body(d);
    }
    if (rc == 0) {
        for (auto &s : a)
            if (s.out && s.bytes && hipMemcpyAsync(s.out, s.dev, s.bytes, hipMemcpyDeviceToHost, stream()) != hipSuccess) {
                set_error("host staging: download failed");
                rc = 1;
            }
        if (hipStreamSynchronize(stream()) != hipSuccess && rc == 0) {
            set_error("host staging: synchronize failed");
            rc = 1;
        }
    } else {
        (void)hipStreamSynchronize(stream());
    }
    return rc;
}

inline size_t fb(int64_t rows, int64_t cols) { return sizeof(float) * (size_t)rows * (size_t)cols; }

} // namespace

namespace amp {
void host_pool_release()
{
    for (int i = 0; i < kMaxStaged; ++i) {
        if (g_pool[i]) (void)hipFree(g_pool[i]);
        g_pool[i] = nullptr;
        g_pool_bytes[i] = 0;
    }
}
} // namespace amp

extern "C" {

int athena_mp_gemm_dw_host(int64_t N, int32_t Fi, int32_t Fo, const float *P, const float *dZ, float *dW)
{
    AMP_REQUIRE(N >= 0 && Fi > 0 && Fo > 0 && P && dZ && dW, "gemm_dw_host: bad arguments");
    return staged({{P, nullptr, fb(N, Fi)}, {dZ, nullptr, fb(N, Fo)}, {nullptr, dW, fb(Fi, Fo)}}, [&](std::vector<void *> &d) {
        return athena_mp_gemm_dw(N, Fi, Fo, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}
int athena_mp_gemm_dx_host(int64_t N, int32_t Fi, int32_t Fo, const float *dZ, const float *W, float *dP)
{
    AMP_REQUIRE(N >= 0 && Fi > 0 && Fo > 0 && dZ && W && dP, "gemm_dx_host: bad arguments");
    return staged({{dZ, nullptr, fb(N, Fo)}, {W, nullptr, fb(Fi, Fo)}, {nullptr, dP, fb(N, Fi)}}, [&](std::vector<void *> &d) {
        return athena_mp_gemm_dx(N, Fi, Fo, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}
int athena_mp_activation_fwd_host(int32_t act, int64_t n, const float *z, float *y)
{
    AMP_REQUIRE(n >= 0 && z && y, "activation_fwd_host: bad arguments");
    return staged({{z, nullptr, fb(n, 1)}, {nullptr, y, fb(n, 1)}}, [&](std::vector<void *> &d) {
        return athena_mp_activation_fwd(act, n, (float *)d[0], (float *)d[1]);
    });
}
int athena_mp_activation_bwd_host(int32_t act, int64_t n, const float *y, const float *g, float *dz)
{
    AMP_REQUIRE(n >= 0 && y && g && dz, "activation_bwd_host: bad arguments");
    return staged({{y, nullptr, fb(n, 1)}, {g, nullptr, fb(n, 1)}, {nullptr, dz, fb(n, 1)}}, [&](std::vector<void *> &d) {
        return athena_mp_activation_bwd(act, n, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}

int athena_mp_duvenaud_propagate_fwd_host(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *x,
                                          const float *e, float *c)
{
    AMP_REQUIRE(g && Fv > 0 && Fe >= 0 && x && c && (Fe == 0 || e), "duvenaud_propagate_fwd_host: bad arguments");
    return staged({{x, nullptr, fb(g->n_cols, Fv)}, {e, nullptr, fb(g->n_edge_cols, Fe)}, {nullptr, c, fb(g->n_rows, Fv + Fe)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_propagate_fwd(g, Fv, Fe, (float *)d[0], Fe ? (float *)d[1] : nullptr, (float *)d[2]);
                  });
}
int athena_mp_duvenaud_propagate_bwd_x_host(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *grad, float *dx)
{
    AMP_REQUIRE(g && Fv > 0 && Fe >= 0 && grad && dx, "duvenaud_propagate_bwd_x_host: bad arguments");
    return staged({{grad, nullptr, fb(g->n_rows, Fv + Fe)}, {nullptr, dx, fb(g->n_cols, Fv)}}, [&](std::vector<void *> &d) {
        return athena_mp_duvenaud_propagate_bwd_x(g, Fv, Fe, (float *)d[0], (float *)d[1]);
    });
}
int athena_mp_duvenaud_propagate_bwd_e_host(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *grad, float *de)
{
    AMP_REQUIRE(g && Fv > 0 && Fe > 0 && grad && de, "duvenaud_propagate_bwd_e_host: bad arguments");
    return staged({{grad, nullptr, fb(g->n_rows, Fv + Fe)}, {nullptr, de, fb(g->n_edge_cols, Fe)}}, [&](std::vector<void *> &d) {
        return athena_mp_duvenaud_propagate_bwd_e(g, Fv, Fe, (float *)d[0], (float *)d[1]);
    });
}
int athena_mp_duvenaud_update_fwd_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                       const float *a, const float *w, float *c)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0 && mx >= mn && a && w && c, "duvenaud_update_fwd_host: bad arguments");
    return staged({{a, nullptr, fb(g->n_rows, Fi)}, {w, nullptr, fb((int64_t)Fi * Fo, mx - mn + 1)}, {nullptr, c, fb(g->n_rows, Fo)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_update_fwd(g, Fi, Fo, mn, mx, (float *)d[0], (float *)d[1], (float *)d[2]);
                  });
}
int athena_mp_duvenaud_update_bwd_a_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                         const float *grad, const float *w, float *da)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0 && mx >= mn && grad && w && da, "duvenaud_update_bwd_a_host: bad arguments");
    return staged({{grad, nullptr, fb(g->n_rows, Fo)}, {w, nullptr, fb((int64_t)Fi * Fo, mx - mn + 1)}, {nullptr, da, fb(g->n_rows, Fi)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_update_bwd_a(g, Fi, Fo, mn, mx, (float *)d[0], (float *)d[1], (float *)d[2]);
                  });
}
int athena_mp_duvenaud_update_bwd_w_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                         const float *grad, const float *a, float *dw)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0 && mx >= mn && grad && a && dw, "duvenaud_update_bwd_w_host: bad arguments");
    return staged({{grad, nullptr, fb(g->n_rows, Fo)}, {a, nullptr, fb(g->n_rows, Fi)}, {nullptr, dw, fb((int64_t)Fi * Fo, mx - mn + 1)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_update_bwd_w(g, Fi, Fo, mn, mx, (float *)d[0], (float *)d[1], (float *)d[2]);
                  });
}
int athena_mp_softmax_segsum_fwd_host(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *logits,
                                      float *p, float *out, int32_t accumulate)
{
    AMP_REQUIRE(O > 0 && N >= 0 && S >= 0 && seg && logits && p && out, "softmax_segsum_fwd_host: bad arguments");
    return staged({{seg, nullptr, sizeof(int32_t) * (size_t)(S + 1)}, {logits, nullptr, fb(N, O)}, {nullptr, p, fb(N, O)},
                   {accumulate ? out : nullptr, out, fb(S, O)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_softmax_segsum_fwd(O, N, S, (int32_t *)d[0], (float *)d[1], (float *)d[2], (float *)d[3], accumulate);
                  });
}
int athena_mp_softmax_segsum_bwd_host(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *p,
                                      const float *gout, float *dlogits)
{
    AMP_REQUIRE(O > 0 && N >= 0 && S > 0 && seg && p && gout && dlogits, "softmax_segsum_bwd_host: bad arguments");
    return staged({{seg, nullptr, sizeof(int32_t) * (size_t)(S + 1)}, {p, nullptr, fb(N, O)}, {gout, nullptr, fb(S, O)},
                   {nullptr, dlogits, fb(N, O)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_softmax_segsum_bwd(O, N, S, (int32_t *)d[0], (float *)d[1], (float *)d[2], (float *)d[3]);
                  });
}

static inline size_t gno_theta_floats(int d, int H, int Fi, int Fo) { return (size_t)H * d + H + (size_t)Fo * Fi * H + (size_t)Fo * Fi; }

int athena_mp_gno_aggregate_fwd_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                     const float *theta, const float *coords, const float *x, float *m)
{
    AMP_REQUIRE(g && d > 0 && H > 0 && Fi > 0 && Fo > 0 && theta && coords && x && m, "gno_aggregate_fwd_host: bad arguments");
    return staged({{theta, nullptr, sizeof(float) * gno_theta_floats(d, H, Fi, Fo)}, {coords, nullptr, fb(g->n_edge_cols, d)},
                   {x, nullptr, fb(g->n_cols, Fi)}, {nullptr, m, fb(g->n_rows, Fo)}},
                  [&](std::vector<void *> &p) {
                      return athena_mp_gno_aggregate_fwd(g, d, H, Fi, Fo, (float *)p[0], (float *)p[1], (float *)p[2], (float *)p[3]);
                  });
}
int athena_mp_gno_aggregate_bwd_x_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                       const float *theta, const float *coords, const float *grad, float *dx)
{
    AMP_REQUIRE(g && d > 0 && H > 0 && Fi > 0 && Fo > 0 && theta && coords && grad && dx, "gno_aggregate_bwd_x_host: bad arguments");
    return staged({{theta, nullptr, sizeof(float) * gno_theta_floats(d, H, Fi, Fo)}, {coords, nullptr, fb(g->n_edge_cols, d)},
                   {grad, nullptr, fb(g->n_rows, Fo)}, {nullptr, dx, fb(g->n_cols, Fi)}},
                  [&](std::vector<void *> &p) {
                      return athena_mp_gno_aggregate_bwd_x(g, d, H, Fi, Fo, (float *)p[0], (float *)p[1], (float *)p[2], (float *)p[3]);
                  });
}
int athena_mp_gno_aggregate_bwd_theta_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                           const float *theta, const float *coords, const float *x, const float *grad,
                                           float *dtheta)
{
    AMP_REQUIRE(g && d > 0 && H > 0 && Fi > 0 && Fo > 0 && theta && coords && x && grad && dtheta,
                "gno_aggregate_bwd_theta_host: bad arguments");
    const size_t tb = sizeof(float) * gno_theta_floats(d, H, Fi, Fo);
    return staged({{theta, nullptr, tb}, {coords, nullptr, fb(g->n_edge_cols, d)}, {x, nullptr, fb(g->n_cols, Fi)},
                   {grad, nullptr, fb(g->n_rows, Fo)}, {nullptr, dtheta, tb}},
                  [&](std::vector<void *> &p) {
                      return athena_mp_gno_aggregate_bwd_theta(g, d, H, Fi, Fo, (float *)p[0], (float *)p[1], (float *)p[2],
                                                               (float *)p[3], (float *)p[4]);
                  });
}
int athena_mp_gno_aggregate_bwd_coords_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                            const float *theta, const float *coords, const float *x, const float *grad,
                                            float *dcoords)
{
    AMP_REQUIRE(g && d > 0 && H > 0 && Fi > 0 && Fo > 0 && theta && coords && x && grad && dcoords,
                "gno_aggregate_bwd_coords_host: bad arguments");
    return staged({{theta, nullptr, sizeof(float) * gno_theta_floats(d, H, Fi, Fo)}, {coords, nullptr, fb(g->n_edge_cols, d)},
                   {x, nullptr, fb(g->n_cols, Fi)}, {grad, nullptr, fb(g->n_rows, Fo)}, {nullptr, dcoords, fb(g->n_edge_cols, d)}},
                  [&](std::vector<void *> &p) {
                      return athena_mp_gno_aggregate_bwd_coords(g, d, H, Fi, Fo, (float *)p[0], (float *)p[1], (float *)p[2],
                                                                (float *)p[3], (float *)p[4]);
                  });
}


// ---- composites and shaped activations ---------------------------------------------------------------
int athena_mp_duvenaud_update_act_fwd_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                           const float *a, const float *w, int32_t act, float *z)
{
    AMP_REQUIRE(g && a && w && z && Fi > 0 && Fo > 0 && mx >= mn, "duvenaud_update_act_fwd_host: bad arguments");
    const int64_t N = g->n_rows;
    return staged({{a, nullptr, fb(N, Fi)}, {w, nullptr, fb((int64_t)Fi * Fo, mx - mn + 1)}, {nullptr, z, fb(N, Fo)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_update_act_fwd(g, Fi, Fo, mn, mx, (float *)d[0], (float *)d[1], act, (float *)d[2]);
                  });
}
int athena_mp_duvenaud_readout_fwd_host(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg, const float *z,
                                        const float *R, float *p, float *out, int32_t accumulate)
{
    AMP_REQUIRE(N >= 0 && Fv > 0 && O > 0 && S >= 0 && seg && z && R && p && out, "duvenaud_readout_fwd_host: bad arguments");
    return staged({{seg, nullptr, sizeof(int32_t) * (size_t)(S + 1)}, {z, nullptr, fb(N, Fv)}, {R, nullptr, fb(O, Fv)},
                   {nullptr, p, fb(N, O)}, {accumulate ? out : nullptr, out, fb(S, O)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_readout_fwd(N, Fv, O, S, (int32_t *)d[0], (float *)d[1], (float *)d[2],
                                                            (float *)d[3], (float *)d[4], accumulate);
                  });
}
int athena_mp_duvenaud_readout_bwd_host(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg, const float *z,
                                        const float *R, const float *p, const float *gout, const float *dz_next,
                                        int32_t act, float *dc, float *dR, int32_t accumulate)
{
    AMP_REQUIRE(N >= 0 && Fv > 0 && O > 0 && S > 0 && seg && z && R && p && gout && dc && dR,
                "duvenaud_readout_bwd_host: bad arguments");
    return staged({{seg, nullptr, sizeof(int32_t) * (size_t)(S + 1)}, {z, nullptr, fb(N, Fv)}, {R, nullptr, fb(O, Fv)},
                   {p, nullptr, fb(N, O)}, {gout, nullptr, fb(S, O)}, {dz_next, nullptr, dz_next ? fb(N, Fv) : 0},
                   {nullptr, dc, fb(N, Fv)}, {accumulate ? dR : nullptr, dR, fb(O, Fv)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_readout_bwd(N, Fv, O, S, (int32_t *)d[0], (float *)d[1], (float *)d[2],
                                                            (float *)d[3], (float *)d[4], dz_next ? (float *)d[5] : nullptr,
                                                            act, (float *)d[6], (float *)d[7], accumulate);
                  });
}
int athena_mp_softmax_fwd_host(int64_t N, int32_t F, const float *z, float *y)
{
    AMP_REQUIRE(N >= 0 && F > 0 && (N == 0 || (z && y)), "softmax_fwd_host: bad arguments");
    return staged({{z, nullptr, fb(N, F)}, {nullptr, y, fb(N, F)}},
                  [&](std::vector<void *> &d) { return athena_mp_softmax_fwd(N, F, (float *)d[0], (float *)d[1]); });
}
int athena_mp_softmax_bwd_host(int64_t N, int32_t F, const float *y, const float *g, float *dz)
{
    AMP_REQUIRE(N >= 0 && F > 0 && (N == 0 || (y && g && dz)), "softmax_bwd_host: bad arguments");
    return staged({{y, nullptr, fb(N, F)}, {g, nullptr, fb(N, F)}, {nullptr, dz, fb(N, F)}}, [&](std::vector<void *> &d) {
        return athena_mp_softmax_bwd(N, F, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}
int athena_mp_swish_fwd_host(int64_t n, float beta, const float *x, float *y)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && y)), "swish_fwd_host: bad arguments");
    return staged({{x, nullptr, fb(n, 1)}, {nullptr, y, fb(n, 1)}},
                  [&](std::vector<void *> &d) { return athena_mp_swish_fwd(n, beta, (float *)d[0], (float *)d[1]); });
}
int athena_mp_swish_bwd_host(int64_t n, float beta, const float *x, const float *g, float *dx)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && g && dx)), "swish_bwd_host: bad arguments");
    return staged({{x, nullptr, fb(n, 1)}, {g, nullptr, fb(n, 1)}, {nullptr, dx, fb(n, 1)}}, [&](std::vector<void *> &d) {
        return athena_mp_swish_bwd(n, beta, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}

int athena_mp_activation_param_fwd_host(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x, float *y)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && y)), "activation_param_fwd_host: bad arguments");
    return staged({{x, nullptr, fb(n, 1)}, {nullptr, y, fb(n, 1)}}, [&](std::vector<void *> &d) {
        return athena_mp_activation_param_fwd(kind, n, scale, p0, p1, (float *)d[0], (float *)d[1]);
    });
}
int athena_mp_activation_param_bwd_host(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x,
                                        const float *g, float *dx)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && g && dx)), "activation_param_bwd_host: bad arguments");
    return staged({{x, nullptr, fb(n, 1)}, {g, nullptr, fb(n, 1)}, {nullptr, dx, fb(n, 1)}}, [&](std::vector<void *> &d) {
        return athena_mp_activation_param_bwd(kind, n, scale, p0, p1, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}

} // extern "C"
