/*
 * diner_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, scalar, loop-structured CPU restatement of the reference DINER render path
 * (tancredeguillou/diner), used ONLY as the checker in tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The shipped path (diner_amd/csrc, HIP for gfx950) never
 * includes, links or calls anything in this file.
 *
 * Parity status: the reference holds NO tests or golden vectors for this path,
 * so this restatement is pinned against outputs of the reference itself, run on the CPU in the
 * build container (oracle/ref_harness.py, oracle/gen_golden.py -> the .npz fixtures in tests/golden,
 * tests/test_oracle_golden.py).
 *
 * Every function cites the reference lines it follows (paths relative to the reference root).
 * All arithmetic is IEEE binary32 with one rounding per reference op (compile with
 * -ffp-contract=off; fused multiply-adds appear only where written as fmaf()).
 *
 * Layouts are the reference's own (single scene, SB = 1; callers loop over SB):
 *   poses [NV,4,4] world->cam, focal/c [NV,2], depths/depths_std [NV,H,W], normals [NV,3,H,W],
 *   latent [NV,C,h,w] (NCHW), rays [NR,8] = o(3) d(3) near far, Linear weights [out,in].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_D_LATENT 512
#define ORC_D_HIDDEN 512
#define ORC_D_IN 55
#define ORC_D_OUT 4
#define ORC_N_BLOCKS 5
#define ORC_COMBINE 3
#define ORC_NFREQ 6
#define ORC_PAD 100 /* exponential padding of the sigma map: image_encoder.py:173 pad_size=100 */

typedef struct {
    int32_t NV, H, W, h, w, C;
    const float *poses, *focal, *c;
    float image_w, image_h;
    const float *depths, *depths_std, *normals, *latent;
    float feature_padding;
    float freq_factor; /* 6.28 in every shipped config (configs/train_diner_facescape.yaml:51) */
} OrcScene;

typedef struct { /* reference nn.Linear layouts, weight [out,in] */
    const float *lin_in_w, *lin_in_b;
    const float *lin_z_w[ORC_COMBINE], *lin_z_b[ORC_COMBINE];
    const float *fc0_w[ORC_N_BLOCKS], *fc0_b[ORC_N_BLOCKS];
    const float *fc1_w[ORC_N_BLOCKS], *fc1_b[ORC_N_BLOCKS];
    const float *lin_out_w, *lin_out_b;
} OrcMlp;

/* ------------------------------------------------------------------------------------ */
/* A.1 stratified candidates -- src/models/nerf_renderer.py:39-63                        */
/* ------------------------------------------------------------------------------------ */
static inline float orc_linspace(int j, int n, float end) {
    /* torch.linspace(0, end, n) in fp32: symmetric evaluation from both ends
       (ATen RangeFactories linspace kernel; scalar/CUDA form) */
    float step = end / (float)(n - 1);
    if (n == 1) return 0.0f;
    if (j < n / 2) return 0.0f + step * (float)j;
    return end - step * (float)(n - j - 1);
}

void orc_sample_coarse(const float *rays, int64_t NR, int NC, const float *u_coarse, float *z)
{
    const double stepd = 1.0 / (double)NC;
    const float end = (float)(1.0 - stepd), step = (float)stepd;
    for (int64_t r = 0; r < NR; ++r) {
        const float near = rays[r * 8 + 6], far = rays[r * 8 + 7];
        for (int j = 0; j < NC; ++j) {
            float t = orc_linspace(j, NC, end);
            t = t + u_coarse[r * NC + j] * step;         /* :57 z_steps += rand * step */
            z[r * NC + j] = near * (1.0f - t) + far * t; /* :60 */
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* A.2 projection -- nerf_renderer.py:99-110, pixelnerf.py:91-108                        */
/* ------------------------------------------------------------------------------------ */
static inline void orc_rot(const float *P, const float *x, float *o)
{ /* row i of the 3x3 block times x; torch.matmul -> BLAS sgemm: k-ordered FMA chain */
    for (int i = 0; i < 3; ++i)
        o[i] = fmaf(P[i * 4 + 2], x[2], fmaf(P[i * 4 + 1], x[1], P[i * 4 + 0] * x[0]));
}

static inline void orc_project(const OrcScene *s, int v, const float *xyz, float *pc, float *uv)
{
    const float *P = s->poses + v * 16;
    orc_rot(P, xyz, pc);
    pc[0] += P[3]; pc[1] += P[7]; pc[2] += P[11];   /* + translation (:101) */
    float u = pc[0] / pc[2], w = pc[1] / pc[2];     /* :107 */
    u = u * s->focal[v * 2 + 0]; w = w * s->focal[v * 2 + 1];
    u = u + s->c[v * 2 + 0];     w = w + s->c[v * 2 + 1];
    uv[0] = u / s->image_w * 2.0f - 1.0f;           /* :110 pixel edges map to -1/+1 */
    uv[1] = w / s->image_h * 2.0f - 1.0f;
}

/* ------------------------------------------------------------------------------------ */
/* A.3 nearest look-ups (F.grid_sample, align_corners=False)                             */
/* ------------------------------------------------------------------------------------ */
/* grid_sampler unnormalize ((u+1)*size-1)/2; ATen's CPU and CUDA kernels are both compiled with
   FMA contraction, i.e. one rounding: fma(u+1, size/2, -0.5) (pinned bit-exact by the goldens) */
static inline float orc_unnorm(float u, int size) { return fmaf(u + 1.0f, (float)size / 2.0f, -0.5f); }
static inline float orc_clipf(float x, int size)
{ /* clamp_max(size-1, clamp_min(0, x)); NaN -> 0 like ATen's operand order */
    float y = (x > 0.0f) ? x : 0.0f;
    return (y < (float)(size - 1)) ? y : (float)(size - 1);
}

/* image_encoder.py:129-151 -- depth: nearest, border */
static inline float orc_depth_nearest(const OrcScene *s, int v, const float *uv)
{
    float ix = nearbyintf(orc_clipf(orc_unnorm(uv[0], s->W), s->W));
    float iy = nearbyintf(orc_clipf(orc_unnorm(uv[1], s->H), s->H));
    return s->depths[((int64_t)v * s->H + (int)iy) * s->W + (int)ix];
}

/* image_encoder.py:182-204 -- normals: nearest, zeros */
static inline void orc_normal_nearest(const OrcScene *s, int v, const float *uv, float *n)
{
    float ix = nearbyintf(orc_unnorm(uv[0], s->W)), iy = nearbyintf(orc_unnorm(uv[1], s->H));
    n[0] = n[1] = n[2] = 0.0f;
    if (!(ix >= 0.0f && ix <= (float)(s->W - 1) && iy >= 0.0f && iy <= (float)(s->H - 1))) return;
    int64_t o = (int64_t)v * 3 * s->H * s->W + (int64_t)(int)iy * s->W + (int)ix;
    n[0] = s->normals[o]; n[1] = s->normals[o + (int64_t)s->H * s->W];
    n[2] = s->normals[o + 2 * (int64_t)s->H * s->W];
}

/* image_encoder.py:153-180 + util/torch_helpers.py:100-160 -- sigma_depth: nearest look-up in
   the map extended by 100 px whose ring at Chebyshev distance d >= 1 holds the replicated border
   value times exp((d-1)/12 * ln 2); zeros beyond. */
static inline float orc_sigma_nearest(const OrcScene *s, int v, const float *uv)
{
    const int Wp = s->W + 2 * ORC_PAD, Hp = s->H + 2 * ORC_PAD;
    float sfx = (float)s->W / (float)Wp, sfy = (float)s->H / (float)Hp; /* torch_helpers.py:158 */
    float jx = nearbyintf(orc_unnorm(uv[0] * sfx, Wp)), jy = nearbyintf(orc_unnorm(uv[1] * sfy, Hp));
    if (!(jx >= 0.0f && jx <= (float)(Wp - 1) && jy >= 0.0f && jy <= (float)(Hp - 1))) return 0.0f;
    int x = (int)jx - ORC_PAD, y = (int)jy - ORC_PAD;
    int dx = x < 0 ? -x : (x > s->W - 1 ? x - (s->W - 1) : 0);
    int dy = y < 0 ? -y : (y > s->H - 1 ? y - (s->H - 1) : 0);
    int cheb = dx > dy ? dx : dy;
    int cx = x < 0 ? 0 : (x > s->W - 1 ? s->W - 1 : x), cy = y < 0 ? 0 : (y > s->H - 1 ? s->H - 1 : y);
    float base = s->depths_std[((int64_t)v * s->H + cy) * s->W + cx];
    float e = (float)(cheb > 1 ? cheb - 1 : 0);
    e = e / 12.0f * (float)0.6931471805599453; /* exponents / double_width * np.log(2) (:121) */
    return base * expf(e);
}

/* ------------------------------------------------------------------------------------ */
/* A.4 surface likelihood of the candidates -- nerf_renderer.py:94-132                   */
/* ------------------------------------------------------------------------------------ */
static void orc_ray_likelihood(const OrcScene *s, const float *ray, const float *z, int NC, float *L)
{
    const float near = ray[6], far = ray[7];
    const float step = (far - near) / (float)NC;          /* :95 */
    const float half = step / 2.0f;
    const float sqrt2 = (float)1.4142135623730951;
    for (int j = 0; j < NC; ++j) L[j] = 0.0f;
    for (int v = 0; v < s->NV; ++v) {
        float dc[3];
        orc_rot(s->poses + v * 16, ray + 3, dc);          /* :103 raydirs_cam */
        for (int j = 0; j < NC; ++j) {
            float xyz[3], pc[3], uv[2], n[3];
            for (int i = 0; i < 3; ++i) xyz[i] = ray[i] + z[j] * ray[3 + i]; /* :96 */
            orc_project(s, v, xyz, pc, uv);
            float d = orc_depth_nearest(s, v, uv);
            float sd = orc_sigma_nearest(s, v, uv);
            orc_normal_nearest(s, v, uv, n);
            float cosd = dc[0] * n[0] + dc[1] * n[1] + dc[2] * n[2];   /* :119 */
            int ok = (cosd <= 0.0f) && (fabsf(d - pc[2]) < 0.05f) && (sd != 0.0f); /* :121-124 */
            if (!ok) continue;
            float den = sd * sqrt2;
            float a = erff((pc[2] + half - d) / den), b = erff((pc[2] - half - d) / den);
            float l = fabsf(0.5f * (a - b));               /* :125-128 */
            if (l > L[j]) L[j] = l;                        /* :129 max over views */
        }
    }
}

void orc_likelihood(const OrcScene *s, const float *rays, int64_t NR, int NC, const float *z_cand,
                    float *L)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t r = 0; r < NR; ++r) orc_ray_likelihood(s, rays + r * 8, z_cand + r * NC, NC, L + r * NC);
}

/* ------------------------------------------------------------------------------------ */
/* A.5 + A.6 short-list and gaussian samples -- nerf_renderer.py:131-132,172-190,        */
/*                                              util/torch_helpers.py:294-302            */
/* n_gauss is DENSE [NR,G]: row r is used iff ray r has a non-zero likelihood.            */
/* Output z_out [NR,K]: slots 0..K-G-1 = the K-G most likely candidates in descending      */
/* likelihood (ties: lower candidate index first; 0 where the likelihood is 0),           */
/* slots K-G..K-1 = gaussian draws (0 for rays without a hit).                            */
/* ------------------------------------------------------------------------------------ */
typedef struct { float l; int idx; } OrcKey;
static int orc_key_cmp(const void *a, const void *b)
{
    const OrcKey *x = (const OrcKey *)a, *y = (const OrcKey *)b;
    if (x->l > y->l) return -1;
    if (x->l < y->l) return 1;
    return x->idx - y->idx;
}

void orc_sample_depthguided(const OrcScene *s, const float *rays, int64_t NR, int NC, int K, int G,
                            const float *z_cand, const float *n_gauss, float *z_out, float *L_out)
{
#pragma omp parallel
    {
        float *L = (float *)malloc(sizeof(float) * NC), *O = (float *)malloc(sizeof(float) * NC);
        OrcKey *keys = (OrcKey *)malloc(sizeof(OrcKey) * NC);
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < NR; ++r) {
            const float *z = z_cand + r * NC;
            float *zo = z_out + r * K;
            orc_ray_likelihood(s, rays + r * 8, z, NC, L);
            if (L_out) memcpy(L_out + r * NC, L, sizeof(float) * NC);
            /* :131-132 occlusion-aware likelihood O_j = L_j * prod_{i<j}(1 - L_i) */
            float cp = 1.0f;
            int hit = 0;
            for (int j = 0; j < NC; ++j) {
                O[j] = (j == 0) ? L[0] : L[j] * cp;
                cp = cp * (1.0f - L[j]);
                hit |= (O[j] != 0.0f);
            }
            /* :172-178 top-K by likelihood, z = 0 where the likelihood is 0 */
            for (int j = 0; j < NC; ++j) { keys[j].l = L[j]; keys[j].idx = j; }
            qsort(keys, NC, sizeof(OrcKey), orc_key_cmp);
            for (int k = 0; k < K; ++k) zo[k] = (keys[k].l == 0.0f) ? 0.0f : z[keys[k].idx];
            if (G > 0) { /* :181-190 */
                if (hit) {
                    float wsum = 0.0f, mean = 0.0f, var = 0.0f;
                    for (int j = 0; j < NC; ++j) wsum += O[j];
                    for (int j = 0; j < NC; ++j) mean += z[j] * (O[j] / wsum);
                    for (int j = 0; j < NC; ++j) { float d = z[j] - mean; var += d * d * (O[j] / wsum); }
                    float sd = sqrtf(var);
                    for (int g = 0; g < G; ++g) zo[K - G + g] = n_gauss[r * G + g] * sd + mean;
                } else {
                    for (int g = 0; g < G; ++g) zo[K - G + g] = 0.0f;
                }
            }
        }
        free(L); free(O); free(keys);
    }
}

/* ------------------------------------------------------------------------------------ */
/* A.7 uniform fill-up of the empty (== 0) slots -- nerf_renderer.py:367-397             */
/* u_fill is DENSE [NR,K]: column i feeds the i-th empty slot (ascending order) of ray r. */
/* ------------------------------------------------------------------------------------ */
static int orc_f_cmp(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

void orc_fill_up(const float *rays, int64_t NR, int K, const float *z_in, const float *u_fill, float *z_out)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < NR; ++r) {
        float *z = z_out + r * K;
        const float near = rays[r * 8 + 6], far = rays[r * 8 + 7];
        memcpy(z, z_in + r * K, sizeof(float) * K);
        qsort(z, K, sizeof(float), orc_f_cmp);            /* :377 */
        int m = 0;
        for (int k = 0; k < K; ++k) m += (z[k] == 0.0f);  /* :380-382 */
        if (m > 0) {
            float step = (far - near) / (float)m;          /* :388 */
            int i = 0;
            for (int k = 0; k < K; ++k) {
                if (z[k] != 0.0f) continue;
                float zm = near + (float)k * step;         /* :389 column index, not rank */
                z[k] = zm + u_fill[r * K + i] * step;      /* :390 */
                ++i;
            }
            qsort(z, K, sizeof(float), orc_f_cmp);        /* :396 */
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* A.8/A.9 per-point, per-view MLP input -- pixelnerf.py:91-128,                         */
/*         positional_encoding.py:33-53, image_encoder.py:97-127                         */
/* out[567] = [latent 512 | p 3, PE(p) 36 | d_cam 3 | delta 1, PE(delta) 12]              */
/* ------------------------------------------------------------------------------------ */
static inline void orc_pe(const float *x, int d, float freq_factor, float *out)
{ /* out[j*d+i] = sin(phase_j + x_i * f_{j/2}); ATen's addcmul kernel is FMA-contracted: one rounding
     (pinned by the goldens: mul+add differs by up to 6e-5 in the encoded value) */
    const float half_pi = (float)(3.141592653589793 * 0.5);
    for (int j = 0; j < 2 * ORC_NFREQ; ++j) {
        float f = freq_factor * (float)(1 << (j / 2));
        float ph = (j & 1) ? half_pi : 0.0f;
        for (int i = 0; i < d; ++i) out[j * d + i] = sinf(fmaf(x[i], f, ph));
    }
}

static void orc_point_view_input(const OrcScene *s, int v, const float *xyz, const float *dir, float *out)
{
    float pc[3], uv[2], dc[3];
    orc_project(s, v, xyz, pc, uv);
    /* latent: bilinear, border, align_corners=False on uv scaled for the feature padding */
    float sx = ((float)s->w - s->feature_padding * 2.0f) / (float)s->w;   /* image_encoder.py:113-114 */
    float sy = ((float)s->h - s->feature_padding * 2.0f) / (float)s->h;
    float ix = orc_clipf(orc_unnorm(uv[0] * sx, s->w), s->w);
    float iy = orc_clipf(orc_unnorm(uv[1] * sy, s->h), s->h);
    float x0f = floorf(ix), y0f = floorf(iy);
    float wx = ix - x0f, ex = 1.0f - wx, wy = iy - y0f, ey = 1.0f - wy;
    float nw = ey * ex, ne = ey * wx, sw = wy * ex, se = wy * wx;
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    int x1ok = x1 <= s->w - 1, y1ok = y1 <= s->h - 1;
    const int64_t plane = (int64_t)s->h * s->w;
    const float *lat = s->latent + (int64_t)v * s->C * plane;
    for (int ch = 0; ch < s->C; ++ch) {
        const float *p = lat + ch * plane;
        float a = p[(int64_t)y0 * s->w + x0];
        float b = x1ok ? p[(int64_t)y0 * s->w + x1] : 0.0f;
        float c = y1ok ? p[(int64_t)y1 * s->w + x0] : 0.0f;
        float d = (x1ok && y1ok) ? p[(int64_t)y1 * s->w + x1] : 0.0f;
        /* ATen accumulates nw, ne, sw, se in this order with contracted FMAs (bit-exact vs goldens) */
        out[ch] = fmaf(d, se, fmaf(c, sw, fmaf(b, ne, a * nw)));
    }
    float *o = out + s->C;
    o[0] = pc[0]; o[1] = pc[1]; o[2] = pc[2];
    orc_pe(pc, 3, s->freq_factor, o + 3);                 /* pixelnerf.py:96 */
    orc_rot(s->poses + v * 16, dir, dc);                  /* :99-101 */
    o[39] = dc[0]; o[40] = dc[1]; o[41] = dc[2];
    float delta = orc_depth_nearest(s, v, uv) - pc[2];   /* :114-115 */
    o[42] = delta;
    orc_pe(&delta, 1, s->freq_factor, o + 43);
}

void orc_point_inputs(const OrcScene *s, const float *xyz, const float *dirs, int64_t B, float *out)
{ /* out [NV,B,567] as the reference's mlp_input */
    const int D = s->C + ORC_D_IN;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int v = 0; v < s->NV; ++v)
            orc_point_view_input(s, v, xyz + b * 3, dirs + b * 3, out + ((int64_t)v * B + b) * D);
}

/* ------------------------------------------------------------------------------------ */
/* A.10 fusion MLP -- resnetfc.py:129-159 (forward), :61-69 (block), :9-14 (combine)     */
/* Dot products are k-ordered FMA chains starting from the bias (nn.Linear = addmm).      */
/* Weights are consumed TRANSPOSED ([in,out]); see orc_mlp_prepare.                       */
/* ------------------------------------------------------------------------------------ */
#define ORC_RB 8 /* rows per micro-block */

typedef struct {
    float *lin_in, *lin_z[ORC_COMBINE], *fc0[ORC_N_BLOCKS], *fc1[ORC_N_BLOCKS], *lin_out;
    const OrcMlp *src;
} OrcMlpT;

static float *orc_transpose(const float *w, int out, int in)
{
    float *t = (float *)malloc(sizeof(float) * (size_t)out * in);
    for (int o = 0; o < out; ++o) for (int i = 0; i < in; ++i) t[(size_t)i * out + o] = w[(size_t)o * in + i];
    return t;
}

OrcMlpT *orc_mlp_prepare(const OrcMlp *m)
{
    OrcMlpT *t = (OrcMlpT *)calloc(1, sizeof(OrcMlpT));
    t->src = m;
    t->lin_in = orc_transpose(m->lin_in_w, ORC_D_HIDDEN, ORC_D_IN);
    for (int b = 0; b < ORC_COMBINE; ++b) t->lin_z[b] = orc_transpose(m->lin_z_w[b], ORC_D_HIDDEN, ORC_D_LATENT);
    for (int b = 0; b < ORC_N_BLOCKS; ++b) {
        t->fc0[b] = orc_transpose(m->fc0_w[b], ORC_D_HIDDEN, ORC_D_HIDDEN);
        t->fc1[b] = orc_transpose(m->fc1_w[b], ORC_D_HIDDEN, ORC_D_HIDDEN);
    }
    t->lin_out = orc_transpose(m->lin_out_w, ORC_D_OUT, ORC_D_HIDDEN);
    return t;
}

void orc_mlp_free(OrcMlpT *t)
{
    free(t->lin_in); free(t->lin_out);
    for (int b = 0; b < ORC_COMBINE; ++b) free(t->lin_z[b]);
    for (int b = 0; b < ORC_N_BLOCKS; ++b) { free(t->fc0[b]); free(t->fc1[b]); }
    free(t);
}

/* y[r][o] = bias[o] + sum_k act(x[r][k]) * wt[k][o], k ascending, fmaf; R rows at once */
static void orc_linear(const float *x, int ldx, int R, int in, const float *wt, const float *bias,
                       int out, int relu_in, float *y, int ldy)
{
    for (int o0 = 0; o0 < out; o0 += 64) {
        int on = out - o0 < 64 ? out - o0 : 64;
        float acc[ORC_RB][64];
        for (int r = 0; r < R; ++r) for (int o = 0; o < on; ++o) acc[r][o] = bias[o0 + o];
        for (int k = 0; k < in; ++k) {
            const float *w = wt + (size_t)k * out + o0;
            for (int r = 0; r < R; ++r) {
                float a = x[(size_t)r * ldx + k];
                if (relu_in) a = a > 0.0f ? a : 0.0f;
                for (int o = 0; o < on; ++o) acc[r][o] = fmaf(a, w[o], acc[r][o]);
            }
        }
        for (int r = 0; r < R; ++r) for (int o = 0; o < on; ++o) y[(size_t)r * ldy + o0 + o] = acc[r][o];
    }
}

/* one block of R points: in [NV][R][567] (view-major, ld = 567) -> out [R][4] raw lin_out */
static void orc_mlp_block(const OrcMlpT *t, const float *in, int64_t view_stride, int NV, int R, float *out)
{
    const OrcMlp *m = t->src;
    const int Hd = ORC_D_HIDDEN, D = ORC_D_LATENT + ORC_D_IN;
    float *x = (float *)malloc(sizeof(float) * R * Hd), *tz = (float *)malloc(sizeof(float) * R * Hd);
    float *net = (float *)malloc(sizeof(float) * R * Hd), *xm = (float *)calloc((size_t)R * Hd, sizeof(float));
    for (int v = 0; v < NV; ++v) {
        const float *zx = in + v * view_stride;
        orc_linear(zx + ORC_D_LATENT, D, R, ORC_D_IN, t->lin_in, m->lin_in_b, Hd, 0, x, Hd); /* :139 */
        for (int b = 0; b < ORC_COMBINE; ++b) {
            orc_linear(zx, D, R, ORC_D_LATENT, t->lin_z[b], m->lin_z_b[b], Hd, 0, tz, Hd);   /* :152 */
            for (int i = 0; i < R * Hd; ++i) x[i] = x[i] + tz[i];                             /* :153 */
            orc_linear(x, Hd, R, Hd, t->fc0[b], m->fc0_b[b], Hd, 1, net, Hd);                 /* :62 */
            orc_linear(net, Hd, R, Hd, t->fc1[b], m->fc1_b[b], Hd, 1, tz, Hd);                /* :63 */
            for (int i = 0; i < R * Hd; ++i) x[i] = x[i] + tz[i];                             /* :69 */
        }
        for (int i = 0; i < R * Hd; ++i) xm[i] = xm[i] + x[i];      /* :146-149 mean over views */
    }
    for (int i = 0; i < R * Hd; ++i) xm[i] = xm[i] / (float)NV;
    for (int b = ORC_COMBINE; b < ORC_N_BLOCKS; ++b) {
        orc_linear(xm, Hd, R, Hd, t->fc0[b], m->fc0_b[b], Hd, 1, net, Hd);
        orc_linear(net, Hd, R, Hd, t->fc1[b], m->fc1_b[b], Hd, 1, tz, Hd);
        for (int i = 0; i < R * Hd; ++i) xm[i] = xm[i] + tz[i];
    }
    orc_linear(xm, Hd, R, Hd, t->lin_out, m->lin_out_b, ORC_D_OUT, 1, out, ORC_D_OUT);        /* :158 */
    free(x); free(tz); free(net); free(xm);
}

/* in [NV,B,567] -> out [B,4] raw (pre-activation) */
void orc_mlp_forward(const OrcMlpT *t, const float *in, int NV, int64_t B, float *out)
{
    const int D = ORC_D_LATENT + ORC_D_IN;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t b0 = 0; b0 < B; b0 += ORC_RB) {
        int R = (int)(B - b0 < ORC_RB ? B - b0 : ORC_RB);
        orc_mlp_block(t, in + b0 * D, B * D, NV, R, out + b0 * ORC_D_OUT);
    }
}

static inline float orc_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

/* PixelNeRF.forward -- pixelnerf.py:55-145: xyz,dirs [B,3] -> rgbsigma [B,4] */
void orc_points_forward(const OrcScene *s, const OrcMlpT *t, const float *xyz, const float *dirs,
                        int64_t B, float *rgbsigma)
{
    const int D = s->C + ORC_D_IN, NV = s->NV;
#pragma omp parallel
    {
        float *in = (float *)malloc(sizeof(float) * (size_t)NV * ORC_RB * D);
#pragma omp for schedule(dynamic, 4)
        for (int64_t b0 = 0; b0 < B; b0 += ORC_RB) {
            int R = (int)(B - b0 < ORC_RB ? B - b0 : ORC_RB);
            float raw[ORC_RB * ORC_D_OUT];
            for (int v = 0; v < NV; ++v)
                for (int r = 0; r < R; ++r)
                    orc_point_view_input(s, v, xyz + (b0 + r) * 3, dirs + (b0 + r) * 3,
                                         in + ((size_t)v * ORC_RB + r) * D);
            orc_mlp_block(t, in, (int64_t)ORC_RB * D, NV, R, raw);
            for (int r = 0; r < R; ++r) {                  /* :139-143 */
                float *o = rgbsigma + (b0 + r) * 4;
                o[0] = orc_sigmoid(raw[r * 4 + 0]); o[1] = orc_sigmoid(raw[r * 4 + 1]);
                o[2] = orc_sigmoid(raw[r * 4 + 2]);
                o[3] = raw[r * 4 + 3] > 0.0f ? raw[r * 4 + 3] : 0.0f;
            }
        }
        free(in);
    }
}

/* ------------------------------------------------------------------------------------ */
/* A.11 alpha compositing -- nerf_renderer.py:299-301,341-360                             */
/* ------------------------------------------------------------------------------------ */
void orc_composite(const float *rays, const float *z, const float *rgbsigma, int64_t NR, int K,
                   int white_bkgd, float *weights, float *rgb, float *depth)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < NR; ++r) {
        const float far = rays[r * 8 + 7];
        const float *zr = z + r * K, *cr = rgbsigma + r * K * 4;
        float T = 1.0f, acc[3] = {0, 0, 0}, dacc = 0.0f, wsum = 0.0f;
        for (int k = 0; k < K; ++k) {
            float delta = (k + 1 < K) ? zr[k + 1] - zr[k] : far - zr[k];   /* :299-301 */
            float sg = cr[k * 4 + 3] > 0.0f ? cr[k * 4 + 3] : 0.0f;       /* relu again (:344) */
            float alpha = 1.0f - expf(-delta * sg);
            float w = alpha * T;                                           /* :351 */
            T = T * (1.0f - alpha + 1e-10f);                               /* :347-350 */
            if (weights) weights[r * K + k] = w;
            for (int c = 0; c < 3; ++c) acc[c] += w * cr[k * 4 + c];       /* :355 */
            dacc += w * zr[k];                                             /* :356 */
            wsum += w;
        }
        for (int c = 0; c < 3; ++c) rgb[r * 3 + c] = white_bkgd ? acc[c] + 1.0f - wsum : acc[c]; /* :357-360 */
        depth[r] = dacc;
    }
}

/* points = o + z*d, viewdirs = d -- nerf_renderer.py:304-307 */
void orc_ray_points(const float *rays, const float *z, int64_t NR, int K, float *xyz, float *dirs)
{
    for (int64_t r = 0; r < NR; ++r)
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < 3; ++i) {
                xyz[(r * K + k) * 3 + i] = rays[r * 8 + i] + z[r * K + k] * rays[r * 8 + 3 + i];
                dirs[(r * K + k) * 3 + i] = rays[r * 8 + 3 + i];
            }
}

/* NeRFRendererDGS.forward -- nerf_renderer.py:399-424 with dense noise */
void orc_render(const OrcScene *s, const OrcMlpT *t, const float *rays, int64_t NR, int NC, int K, int G,
                int white_bkgd, const float *u_coarse, const float *n_gauss, const float *u_fill,
                float *z_out, float *rgbsigma, float *weights, float *rgb, float *depth)
{
    float *zc = (float *)malloc(sizeof(float) * NR * NC), *zd = (float *)malloc(sizeof(float) * NR * K);
    float *xyz = (float *)malloc(sizeof(float) * NR * K * 3), *dirs = (float *)malloc(sizeof(float) * NR * K * 3);
    orc_sample_coarse(rays, NR, NC, u_coarse, zc);
    orc_sample_depthguided(s, rays, NR, NC, K, G, zc, n_gauss, zd, NULL);
    orc_fill_up(rays, NR, K, zd, u_fill, z_out);
    orc_ray_points(rays, z_out, NR, K, xyz, dirs);
    orc_points_forward(s, t, xyz, dirs, NR * K, rgbsigma);
    orc_composite(rays, z_out, rgbsigma, NR, K, white_bkgd, weights, rgb, depth);
    free(zc); free(zd); free(xyz); free(dirs);
}
