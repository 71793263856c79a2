/*
 * TEST INFRASTRUCTURE ONLY (oracle/).  CPU restatement of the reference Wave-U-Net
 * forward / loss / backward used as the parity checker for the HIP path.  Nothing in
 * the product package may load this library; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * Parity pinning: the reference ships no tests or golden vectors, so this
 * restatement is pinned against outputs of the reference itself, imported in the build
 * container by tests/golden/make_golden.py (fixtures committed under tests/golden/,
 * checked by tests/test_oracle_golden.py).
 *
 * What it restates (all file:line relative to /root/reference):
 *   model/unet_basic.py:6-17   DownSamplingLayer  = Conv1d(k15,s1,p7) -> BatchNorm1d -> LeakyReLU(0.1)
 *   model/unet_basic.py:19-30  UpSamplingLayer    = Conv1d(k5,s1,p2)  -> BatchNorm1d -> LeakyReLU(0.1)
 *   model/unet_basic.py:33-75  channel plan
 *   model/unet_basic.py:77-100 forward: encoder loop + [:, :, ::2] decimation, middle,
 *                              F.interpolate(x2, linear, align_corners=True) + cat([up, skip]) + decoder,
 *                              cat([o, input]) -> Conv1d(25->1, k1) -> Tanh
 *   model/loss.py:3-7          MSELoss / L1Loss (+ SmoothL1Loss(beta=1))
 *   trainer/trainer.py:36-37   loss(clean, enhanced); loss.backward()
 * and the ATen semantics those calls resolve to (torch is a dependency of the reference, not
 * vendored; only prose pin "Pytorch 1.2.0", README.md:27; container has torch 2.10.0):
 *   - conv1d zero padding, cross-correlation orientation;
 *   - batch_norm training: biased variance for normalisation, unbiased into running_var,
 *     momentum 0.1, eps 1e-5, num_batches_tracked += 1;
 *   - upsample_linear1d align_corners=True with *fp32* coordinate arithmetic
 *     (UpSample.h area_pixel_compute_scale / area_pixel_compute_source_index /
 *     guard_index_and_lambda): scale=(float)(Lin-1)/(Lout-1), src=scale*j, i0=(int)src,
 *     l1=src-i0, l0=1-l1, i1=i0+(i0<Lin-1).  The coordinate arithmetic stays in `float`
 *     even when REAL is double - SURVEY.md §7: an exact-coordinate implementation is 1.4e-3 away.
 *
 * Build: gcc -O3 -fopenmp -fPIC -shared -DREAL=float  -o libwunet_oracle_f32.so wunet_oracle.c -lm
 *        gcc -O3 -fopenmp -fPIC -shared -DREAL=double -o libwunet_oracle_f64.so wunet_oracle.c -lm
 * I/O is always float32; REAL is the internal arithmetic type (double = arbiter).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

#define BN_EPS 1e-5
#define BN_MOM 0.1
#define SLOPE 0.1

typedef long long i64;

/* ------------------------------------------------------------------ ops */

/* z[b,co,l] = bias[co] + sum_{ci,k} w[co,ci,k] * x[b,ci,l+k-pad]   (nn.Conv1d, stride 1, pad K/2) */
static void conv1d_fwd(const REAL *x, const REAL *w, const REAL *bias, REAL *z,
                       int B, int Cin, int Cout, int L, int K)
{
    const int pad = K / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            REAL *zr = z + ((size_t)b * Cout + co) * L;
            const REAL bv = bias ? bias[co] : (REAL)0;
            for (int l = 0; l < L; ++l) zr[l] = bv;
            for (int ci = 0; ci < Cin; ++ci) {
                const REAL *xr = x + ((size_t)b * Cin + ci) * L;
                const REAL *wr = w + ((size_t)co * Cin + ci) * K;
                for (int k = 0; k < K; ++k) {
                    const REAL wv = wr[k];
                    const int off = k - pad;
                    int lo = off < 0 ? -off : 0;
                    int hi = off > 0 ? L - off : L;
                    for (int l = lo; l < hi; ++l) zr[l] += wv * xr[l + off];
                }
            }
        }
}

/* dx[b,ci,l] = sum_{co,k} w[co,ci,k] * gz[b,co,l-k+pad] */
static void conv1d_dgrad(const REAL *gz, const REAL *w, REAL *dx,
                         int B, int Cin, int Cout, int L, int K)
{
    const int pad = K / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Cin; ++ci) {
            REAL *dr = dx + ((size_t)b * Cin + ci) * L;
            for (int l = 0; l < L; ++l) dr[l] = 0;
            for (int co = 0; co < Cout; ++co) {
                const REAL *gr = gz + ((size_t)b * Cout + co) * L;
                const REAL *wr = w + ((size_t)co * Cin + ci) * K;
                for (int k = 0; k < K; ++k) {
                    const REAL wv = wr[k];
                    const int off = pad - k;          /* dx[l] += w * gz[l + off] */
                    int lo = off < 0 ? -off : 0;
                    int hi = off > 0 ? L - off : L;
                    for (int l = lo; l < hi; ++l) dr[l] += wv * gr[l + off];
                }
            }
        }
}

/* dw[co,ci,k] = sum_{b,l} gz[b,co,l] * x[b,ci,l+k-pad];  db[co] = sum gz  (double accumulation) */
static void conv1d_wgrad(const REAL *gz, const REAL *x, REAL *dw, REAL *db,
                         int B, int Cin, int Cout, int L, int K)
{
    const int pad = K / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int k = 0; k < K; ++k) {
                const int off = k - pad;
                int lo = off < 0 ? -off : 0;
                int hi = off > 0 ? L - off : L;
                double acc = 0.0;
                for (int b = 0; b < B; ++b) {
                    const REAL *gr = gz + ((size_t)b * Cout + co) * L;
                    const REAL *xr = x + ((size_t)b * Cin + ci) * L;
                    double s = 0.0;
                    for (int l = lo; l < hi; ++l) s += (double)gr[l] * (double)xr[l + off];
                    acc += s;
                }
                dw[((size_t)co * Cin + ci) * K + k] = (REAL)acc;
            }
    if (db) {
#pragma omp parallel for schedule(static)
        for (int co = 0; co < Cout; ++co) {
            double acc = 0.0;
            for (int b = 0; b < B; ++b) {
                const REAL *gr = gz + ((size_t)b * Cout + co) * L;
                for (int l = 0; l < L; ++l) acc += gr[l];
            }
            db[co] = (REAL)acc;
        }
    }
}

/* BatchNorm1d forward.  training: batch stats (biased var), running update with unbiased var.
 * eval: running stats.  Saves mean / rstd actually used.  y may alias z. */
static void bn_fwd(const REAL *z, const REAL *gamma, const REAL *beta, REAL *rmean, REAL *rvar,
                   REAL *save_mean, REAL *save_rstd, REAL *y, int B, int C, int L, int training)
{
    const double n = (double)B * L;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double mean, var;
        if (training) {
            double s = 0.0;
            for (int b = 0; b < B; ++b) {
                const REAL *zr = z + ((size_t)b * C + c) * L;
                for (int l = 0; l < L; ++l) s += zr[l];
            }
            mean = s / n;
            double q = 0.0;
            for (int b = 0; b < B; ++b) {
                const REAL *zr = z + ((size_t)b * C + c) * L;
                for (int l = 0; l < L; ++l) { double d = zr[l] - mean; q += d * d; }
            }
            var = q / n;
            rmean[c] = (REAL)((1.0 - BN_MOM) * rmean[c] + BN_MOM * mean);
            rvar[c] = (REAL)((1.0 - BN_MOM) * rvar[c] + BN_MOM * (n > 1 ? q / (n - 1) : var));
        } else {
            mean = rmean[c];
            var = rvar[c];
        }
        const REAL rstd = (REAL)(1.0 / sqrt(var + BN_EPS));
        const REAL mu = (REAL)mean;
        save_mean[c] = mu;
        save_rstd[c] = rstd;
        const REAL a = gamma[c] * rstd, s0 = beta[c] - mu * a;
        for (int b = 0; b < B; ++b) {
            const REAL *zr = z + ((size_t)b * C + c) * L;
            REAL *yr = y + ((size_t)b * C + c) * L;
            for (int l = 0; l < L; ++l) yr[l] = (zr[l] - mu) * rstd * gamma[c] + beta[c];
            (void)s0;
        }
    }
}

/* BatchNorm1d training backward: gz = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) */
static void bn_bwd(const REAL *g, const REAL *z, const REAL *gamma, const REAL *mean, const REAL *rstd,
                   REAL *gz, REAL *dgamma, REAL *dbeta, int B, int C, int L)
{
    const double n = (double)B * L;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double s1 = 0.0, s2 = 0.0;
        for (int b = 0; b < B; ++b) {
            const REAL *gr = g + ((size_t)b * C + c) * L;
            const REAL *zr = z + ((size_t)b * C + c) * L;
            for (int l = 0; l < L; ++l) {
                double xh = ((double)zr[l] - mean[c]) * rstd[c];
                s1 += gr[l];
                s2 += gr[l] * xh;
            }
        }
        dgamma[c] = (REAL)s2;
        dbeta[c] = (REAL)s1;
        const double m1 = s1 / n, m2 = s2 / n, a = (double)gamma[c] * rstd[c];
        for (int b = 0; b < B; ++b) {
            const REAL *gr = g + ((size_t)b * C + c) * L;
            const REAL *zr = z + ((size_t)b * C + c) * L;
            REAL *or_ = gz + ((size_t)b * C + c) * L;
            for (int l = 0; l < L; ++l) {
                double xh = ((double)zr[l] - mean[c]) * rstd[c];
                or_[l] = (REAL)(a * (gr[l] - m1 - xh * m2));
            }
        }
    }
}

/* ATen upsample_linear1d align_corners=True source index, fp32 arithmetic on purpose. */
static inline void up_coord(int j, int Lin, int Lout, int *i0, int *i1, float *l0, float *l1)
{
    const float scale = Lout > 1 ? (float)(Lin - 1) / (float)(Lout - 1) : 0.0f;
    const float src = scale * (float)j;
    int a = (int)floorf(src);
    if (a > Lin - 1) a = Lin - 1;
    float lam = src - (float)a;
    if (lam < 0.0f) lam = 0.0f;
    if (lam > 1.0f) lam = 1.0f;
    *i0 = a;
    *i1 = a + (a < Lin - 1 ? 1 : 0);
    *l1 = lam;
    *l0 = 1.0f - lam;
}

/* ------------------------------------------------------------------ op-level C ABI (float I/O) */

static REAL *to_real(const float *p, size_t n)
{
    REAL *r = (REAL *)malloc(sizeof(REAL) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) r[i] = (REAL)p[i];
    return r;
}
static void from_real(const REAL *r, float *p, size_t n)
{
    for (size_t i = 0; i < n; ++i) p[i] = (float)r[i];
}

int wuo_real_bytes(void) { return (int)sizeof(REAL); }

int wuo_conv1d_fwd(const float *x, const float *w, const float *bias, float *z,
                   int B, int Cin, int Cout, int L, int K)
{
    REAL *xr = to_real(x, (size_t)B * Cin * L), *wr = to_real(w, (size_t)Cout * Cin * K);
    REAL *br = bias ? to_real(bias, Cout) : NULL;
    REAL *zr = (REAL *)malloc(sizeof(REAL) * (size_t)B * Cout * L);
    conv1d_fwd(xr, wr, br, zr, B, Cin, Cout, L, K);
    from_real(zr, z, (size_t)B * Cout * L);
    free(xr); free(wr); free(br); free(zr);
    return 0;
}

int wuo_conv1d_bwd(const float *gz, const float *x, const float *w, float *dx, float *dw, float *db,
                   int B, int Cin, int Cout, int L, int K)
{
    REAL *gr = to_real(gz, (size_t)B * Cout * L), *xr = to_real(x, (size_t)B * Cin * L);
    REAL *wr = to_real(w, (size_t)Cout * Cin * K);
    if (dx) {
        REAL *d = (REAL *)malloc(sizeof(REAL) * (size_t)B * Cin * L);
        conv1d_dgrad(gr, wr, d, B, Cin, Cout, L, K);
        from_real(d, dx, (size_t)B * Cin * L);
        free(d);
    }
    if (dw) {
        REAL *d = (REAL *)malloc(sizeof(REAL) * (size_t)Cout * Cin * K);
        REAL *b = (REAL *)malloc(sizeof(REAL) * (size_t)Cout);
        conv1d_wgrad(gr, xr, d, b, B, Cin, Cout, L, K);
        from_real(d, dw, (size_t)Cout * Cin * K);
        if (db) from_real(b, db, Cout);
        free(d); free(b);
    }
    free(gr); free(xr); free(wr);
    return 0;
}

/* y[b,c,j] = l0*x[b,c,i0] + l1*x[b,c,i1], Lout = 2*Lin  (F.interpolate, unet_basic.py:93) */
int wuo_upsample2x_fwd(const float *x, float *y, int B, int C, int Lin)
{
    const int Lout = 2 * Lin;
    for (size_t r = 0; r < (size_t)B * C; ++r)
        for (int j = 0; j < Lout; ++j) {
            int i0, i1; float l0, l1;
            up_coord(j, Lin, Lout, &i0, &i1, &l0, &l1);
            y[r * Lout + j] = (float)((REAL)l0 * (REAL)x[r * Lin + i0] + (REAL)l1 * (REAL)x[r * Lin + i1]);
        }
    return 0;
}

int wuo_upsample2x_bwd(const float *gy, float *gx, int B, int C, int Lin)
{
    const int Lout = 2 * Lin;
    for (size_t r = 0; r < (size_t)B * C; ++r) {
        REAL *acc = (REAL *)calloc(Lin, sizeof(REAL));
        for (int j = 0; j < Lout; ++j) {
            int i0, i1; float l0, l1;
            up_coord(j, Lin, Lout, &i0, &i1, &l0, &l1);
            acc[i0] += (REAL)l0 * (REAL)gy[r * Lout + j];
            acc[i1] += (REAL)l1 * (REAL)gy[r * Lout + j];
        }
        for (int i = 0; i < Lin; ++i) gx[r * Lin + i] = (float)acc[i];
        free(acc);
    }
    return 0;
}

/* ------------------------------------------------------------------ whole network */

typedef struct {
    int cin, cout, taps, L;       /* L = length this conv runs at */
    REAL *x, *z, *y;              /* materialised conv input, raw conv output, activated output */
    REAL *mean, *rstd;
} layer_t;

static void lrelu_fwd(const REAL *y, REAL *o, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) o[i] = y[i] > 0 ? y[i] : (REAL)SLOPE * y[i];
}

/* kind: 0 mse, 1 l1, 2 smooth_l1(beta=1); returns loss, writes d loss / d enhanced */
static double loss_fwd_bwd(int kind, const float *clean, const REAL *enh, REAL *g, size_t n)
{
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double d = (double)enh[i] - (double)clean[i];
        double l, gd;
        if (kind == 0) { l = d * d; gd = 2.0 * d; }
        else if (kind == 1) { l = fabs(d); gd = (d > 0) - (d < 0); }
        else { const double a = fabs(d); if (a < 1.0) { l = 0.5 * d * d; gd = d; } else { l = a - 0.5; gd = (d > 0) - (d < 0); } }
        acc += l;
        if (g) g[i] = (REAL)(gd / (double)n);
    }
    return acc / (double)n;
}

/*
 * Full step.  params: 4*(2n+1)+2 float* in canonical order (oracle/plan.py:param_names);
 * running: 2*(2n+1) float* (mean,var per layer; updated in place when training);
 * nbt: (2n+1) int64 (incremented when training).  grads: same shapes/order as params, or NULL
 * for forward only.  clean may be NULL when grads is NULL.  loss_out may be NULL.
 * acts_out (optional): 2n+1 float* receiving each layer's raw conv output z (pre-BN).
 */
int wuo_step(int n_layers, int ci, int B, int T, const float *const *params, float *const *running,
             i64 *nbt, const float *noisy, const float *clean, int training, int loss_kind,
             float *out, float *loss_out, float *const *grads, float *const *acts_out)
{
    const int NL = 2 * n_layers + 1;
    if (T % (1 << n_layers) != 0 || (T >> n_layers) < 1) return -1;
    layer_t *ly = (layer_t *)calloc(NL, sizeof(layer_t));
    for (int i = 0; i < n_layers; ++i) {
        ly[i].cin = i == 0 ? 1 : i * ci; ly[i].cout = (i + 1) * ci; ly[i].taps = 15; ly[i].L = T >> i;
    }
    ly[n_layers].cin = ly[n_layers].cout = n_layers * ci; ly[n_layers].taps = 15; ly[n_layers].L = T >> n_layers;
    for (int j = 0; j < n_layers; ++j) {
        layer_t *d = &ly[n_layers + 1 + j];
        d->cout = (n_layers - j) * ci;
        d->cin = j == 0 ? 2 * n_layers * ci : (2 * (n_layers - j) + 1) * ci;
        d->taps = 5; d->L = T >> (n_layers - 1 - j);
    }
    REAL **P = (REAL **)calloc(4 * NL + 2, sizeof(REAL *));
    REAL **RS = (REAL **)calloc(2 * NL, sizeof(REAL *));
    for (int i = 0; i < NL; ++i) {
        const size_t wn = (size_t)ly[i].cout * ly[i].cin * ly[i].taps;
        P[4 * i] = to_real(params[4 * i], wn);
        for (int q = 1; q < 4; ++q) P[4 * i + q] = to_real(params[4 * i + q], ly[i].cout);
        RS[2 * i] = to_real(running[2 * i], ly[i].cout);
        RS[2 * i + 1] = to_real(running[2 * i + 1], ly[i].cout);
    }
    P[4 * NL] = to_real(params[4 * NL], ci + 1);
    P[4 * NL + 1] = to_real(params[4 * NL + 1], 1);

    REAL *in = to_real(noisy, (size_t)B * T);

    /* ---- forward (unet_basic.py:77-100) */
    const REAL *prev = in; int prevC = 1, prevL = T;
    for (int i = 0; i < NL; ++i) {
        layer_t *l = &ly[i];
        const size_t xn = (size_t)B * l->cin * l->L, zn = (size_t)B * l->cout * l->L;
        l->x = (REAL *)malloc(sizeof(REAL) * xn);
        l->z = (REAL *)malloc(sizeof(REAL) * zn);
        l->y = (REAL *)malloc(sizeof(REAL) * zn);
        l->mean = (REAL *)malloc(sizeof(REAL) * l->cout);
        l->rstd = (REAL *)malloc(sizeof(REAL) * l->cout);
        if (i == 0) {
            memcpy(l->x, in, sizeof(REAL) * xn);
        } else if (i <= n_layers) {
            /* o = o[:, :, ::2]  (unet_basic.py:86) */
            for (size_t r = 0; r < (size_t)B * prevC; ++r)
                for (int t = 0; t < l->L; ++t) l->x[r * l->L + t] = prev[r * prevL + 2 * t];
        } else {
            /* interpolate x2 + cat([up, skip])  (unet_basic.py:93-95) */
            const layer_t *sk = &ly[n_layers - 1 - (i - n_layers - 1)];
            for (int b = 0; b < B; ++b) {
                for (int c = 0; c < prevC; ++c) {
                    const REAL *src = prev + ((size_t)b * prevC + c) * prevL;
                    REAL *dst = l->x + ((size_t)b * l->cin + c) * l->L;
                    for (int j = 0; j < l->L; ++j) {
                        int i0, i1; float l0, l1;
                        up_coord(j, prevL, l->L, &i0, &i1, &l0, &l1);
                        dst[j] = (REAL)l0 * src[i0] + (REAL)l1 * src[i1];
                    }
                }
                for (int c = 0; c < sk->cout; ++c)
                    memcpy(l->x + ((size_t)b * l->cin + prevC + c) * l->L,
                           sk->y + ((size_t)b * sk->cout + c) * sk->L, sizeof(REAL) * l->L);
            }
        }
        conv1d_fwd(l->x, P[4 * i], P[4 * i + 1], l->z, B, l->cin, l->cout, l->L, l->taps);
        bn_fwd(l->z, P[4 * i + 2], P[4 * i + 3], RS[2 * i], RS[2 * i + 1], l->mean, l->rstd, l->y,
               B, l->cout, l->L, training);
        lrelu_fwd(l->y, l->y, zn);
        if (training) nbt[i] += 1;
        if (acts_out && acts_out[i]) from_real(l->z, acts_out[i], zn);
        prev = l->y; prevC = l->cout; prevL = l->L;
    }
    /* head: cat([o, input]) -> 1x1 conv -> tanh  (unet_basic.py:98-99) */
    const layer_t *last = &ly[NL - 1];
    const REAL *hw = P[4 * NL], hb = P[4 * NL + 1][0];
    REAL *o = (REAL *)malloc(sizeof(REAL) * (size_t)B * T);
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            REAL s = hb;
            for (int c = 0; c < ci; ++c) s += hw[c] * last->y[((size_t)b * ci + c) * T + t];
            s += hw[ci] * in[(size_t)b * T + t];
            o[(size_t)b * T + t] = (REAL)tanh((double)s);
        }
    from_real(o, out, (size_t)B * T);

    if (training)
        for (int i = 0; i < NL; ++i) { from_real(RS[2 * i], running[2 * i], ly[i].cout); from_real(RS[2 * i + 1], running[2 * i + 1], ly[i].cout); }

    REAL *go = NULL;
    if (clean) {
        go = (REAL *)malloc(sizeof(REAL) * (size_t)B * T);
        double lv = loss_fwd_bwd(loss_kind, clean, o, grads ? go : NULL, (size_t)B * T);
        if (loss_out) *loss_out = (float)lv;
    }

    /* ---- backward (autograd of the above; trainer/trainer.py:37) */
    if (grads && clean) {
        /* head */
        double dwh[64 * 64]; double dbh = 0.0;   /* ci+1 <= 4096 is plenty */
        for (int c = 0; c <= ci; ++c) dwh[c] = 0.0;
        REAL **gy = (REAL **)calloc(NL, sizeof(REAL *));
        for (int i = 0; i < NL; ++i) gy[i] = (REAL *)calloc((size_t)B * ly[i].cout * ly[i].L, sizeof(REAL));
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t) {
                const size_t p = (size_t)b * T + t;
                const REAL gs = go[p] * ((REAL)1 - o[p] * o[p]);
                dbh += gs;
                for (int c = 0; c < ci; ++c) {
                    const size_t q = ((size_t)b * ci + c) * T + t;
                    dwh[c] += (double)gs * last->y[q];
                    gy[NL - 1][q] = gs * hw[c];
                }
                dwh[ci] += (double)gs * in[p];
            }
        for (int c = 0; c <= ci; ++c) grads[4 * NL][c] = (float)dwh[c];
        grads[4 * NL + 1][0] = (float)dbh;

        for (int i = NL - 1; i >= 0; --i) {
            layer_t *l = &ly[i];
            const size_t zn = (size_t)B * l->cout * l->L, xn = (size_t)B * l->cin * l->L;
            /* LeakyReLU backward on the BN output sign (y>0 <=> pre-activation>0) */
            for (size_t q = 0; q < zn; ++q) if (!(l->y[q] > 0)) gy[i][q] *= (REAL)SLOPE;
            REAL *gz = (REAL *)malloc(sizeof(REAL) * zn);
            REAL *dg = (REAL *)malloc(sizeof(REAL) * l->cout), *dbt = (REAL *)malloc(sizeof(REAL) * l->cout);
            bn_bwd(gy[i], l->z, P[4 * i + 2], l->mean, l->rstd, gz, dg, dbt, B, l->cout, l->L);
            from_real(dg, grads[4 * i + 2], l->cout);
            from_real(dbt, grads[4 * i + 3], l->cout);
            REAL *dw = (REAL *)malloc(sizeof(REAL) * (size_t)l->cout * l->cin * l->taps);
            REAL *db = (REAL *)malloc(sizeof(REAL) * l->cout);
            conv1d_wgrad(gz, l->x, dw, db, B, l->cin, l->cout, l->L, l->taps);
            from_real(dw, grads[4 * i], (size_t)l->cout * l->cin * l->taps);
            from_real(db, grads[4 * i + 1], l->cout);
            free(dw); free(db); free(dg); free(dbt);
            if (i > 0) {
                REAL *dx = (REAL *)malloc(sizeof(REAL) * xn);
                conv1d_dgrad(gz, P[4 * i], dx, B, l->cin, l->cout, l->L, l->taps);
                if (i <= n_layers) {
                    /* input was prev[:, :, ::2]: zero-stuffed scatter-add into the producer's grad */
                    layer_t *pl = &ly[i - 1];
                    for (size_t r = 0; r < (size_t)B * pl->cout; ++r)
                        for (int t = 0; t < l->L; ++t) gy[i - 1][r * pl->L + 2 * t] += dx[r * l->L + t];
                } else {
                    layer_t *pl = &ly[i - 1];
                    const int ski = n_layers - 1 - (i - n_layers - 1);
                    layer_t *sk = &ly[ski];
                    for (int b = 0; b < B; ++b) {
                        for (int c = 0; c < pl->cout; ++c) {
                            const REAL *g = dx + ((size_t)b * l->cin + c) * l->L;
                            REAL *dst = gy[i - 1] + ((size_t)b * pl->cout + c) * pl->L;
                            for (int j = 0; j < l->L; ++j) {
                                int i0, i1; float l0, l1;
                                up_coord(j, pl->L, l->L, &i0, &i1, &l0, &l1);
                                dst[i0] += (REAL)l0 * g[j];
                                dst[i1] += (REAL)l1 * g[j];
                            }
                        }
                        for (int c = 0; c < sk->cout; ++c) {
                            const REAL *g = dx + ((size_t)b * l->cin + pl->cout + c) * l->L;
                            REAL *dst = gy[ski] + ((size_t)b * sk->cout + c) * sk->L;
                            for (int t = 0; t < l->L; ++t) dst[t] += g[t];
                        }
                    }
                }
                free(dx);
            }
            free(gz);
        }
        for (int i = 0; i < NL; ++i) free(gy[i]);
        free(gy);
    }

    for (int i = 0; i < NL; ++i) { free(ly[i].x); free(ly[i].z); free(ly[i].y); free(ly[i].mean); free(ly[i].rstd); }
    for (int i = 0; i < 4 * NL + 2; ++i) free(P[i]);
    for (int i = 0; i < 2 * NL; ++i) free(RS[i]);
    free(P); free(RS); free(ly); free(in); free(o); free(go);
    return 0;
}
