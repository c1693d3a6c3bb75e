/*
 * TEST INFRASTRUCTURE ONLY (oracle).  Not part of the shipped library.
 *
 * Generic body of the CPU restatement of quartiq/stabilizer-stream src/psd.rs.
 * Included twice by psd_oracle.c: REAL=float (mirrors the reference's f32
 * arithmetic statement by statement) and REAL=double (numerical truth).
 *
 * Required macros: REAL, SFX(name), R_SIN, R_IS_F32
 */

/* ---- Window (src/psd.rs:12-56) ------------------------------------------ */

typedef struct {
    int n;
    REAL *win;
    REAL power;   /* src/psd.rs:15 */
    REAL nenbw;   /* src/psd.rs:17 */
    int overlap;  /* src/psd.rs:19 */
} SFX(ora_window_t);

/* kind 0: rectangular (src/psd.rs:24-32); kind 1: hann (src/psd.rs:42-55) */
static int SFX(window_init)(SFX(ora_window_t) * w, int n, int kind)
{
    w->n = n;
    w->win = (REAL *)malloc(sizeof(REAL) * (size_t)n);
    if (!w->win)
        return -1;
    if (kind == 0) {
        for (int i = 0; i < n; ++i)
            w->win[i] = (REAL)1.0;
        w->power = (REAL)1.0;
        w->nenbw = (REAL)1.0;
        w->overlap = 0;
    } else {
#if R_IS_F32
        /* let df = core::f32::consts::PI / N as f32;  (src/psd.rs:44) */
        const float df = 3.14159265358979323846f / (float)n;
        for (int i = 0; i < n; ++i) {
            /* *w = (df * i as f32).sin().powi(2);  (src/psd.rs:47) */
            float s = sinf(df * (float)i);
            w->win[i] = s * s;
        }
#else
        for (int i = 0; i < n; ++i) {
            double s = sin(M_PI * (double)i / (double)n);
            w->win[i] = s * s;
        }
#endif
        w->power = (REAL)0.25;
        w->nenbw = (REAL)1.5;
        w->overlap = n / 2;
    }
    return 0;
}

/* A caller-built Window<N> (pub struct, pub fields, src/psd.rs:12-20): the weights and constants are DEFINED by
 * the caller in f32; the f64 instantiation only widens them. */
static int SFX(window_init_table)(SFX(ora_window_t) * w, int n, const float *win, float power, float nenbw, int overlap)
{
    if (overlap < 0 || overlap >= n)
        return -1;
    w->n = n;
    w->win = (REAL *)malloc(sizeof(REAL) * (size_t)n);
    if (!w->win)
        return -1;
    for (int i = 0; i < n; ++i)
        w->win[i] = (REAL)win[i];
    w->power = (REAL)power;
    w->nenbw = (REAL)nenbw;
    w->overlap = overlap;
    return 0;
}

/* ---- FFT (src/psd.rs:213; rustfft 6.4.1 forward, unnormalised) ---------- */
/* X[k] = sum_j c[j] exp(-2 pi i jk/N).  Iterative radix-2 DIT; any N = 2^m. */

typedef struct {
    int n;
    REAL *tw_re, *tw_im; /* n/2 twiddles exp(-2 pi i k/n) */
    int *rev;
    /* bench.py's cpu_baseline only (SFX(ora_cascade_set_fast_fft)): a radix-4 Stockham plan whose inner loops
     * gcc vectorises -- a fairer stand-in for rustfft's SIMD butterflies than the scalar radix-2 above, which stays
     * the arithmetic every parity test sees */
    int fast, fast_stages;
    REAL *ftw;  /* per stage: 6 arrays of n/4 (w1 re, w1 im, w2 re, w2 im, w3 re, w3 im), by butterfly index */
    REAL *fbuf; /* 4 n: two (re, im) planar buffers */
} SFX(ora_fft);

static int SFX(fft_init)(SFX(ora_fft) * f, int n)
{
    if (n < 1)
        return -1;
    f->n = n;
    f->fast = 0;
    f->fast_stages = 0;
    f->ftw = NULL;
    f->fbuf = NULL;
    f->rev = NULL;
    if (n & (n - 1)) {
        /* Not a power of two (rustfft plans any length, src/psd.rs:418): the DFT by its definition, O(n^2), with the n
         * twiddles exp(-2 pi i k/n) tabulated from f64 -- test infrastructure, exactness before speed */
        f->tw_re = (REAL *)malloc(sizeof(REAL) * (size_t)n);
        f->tw_im = (REAL *)malloc(sizeof(REAL) * (size_t)n);
        f->fbuf = (REAL *)malloc(sizeof(REAL) * 2 * (size_t)n);
        if (!f->tw_re || !f->tw_im || !f->fbuf)
            return -1;
        for (int k = 0; k < n; ++k) {
            double a = -2.0 * M_PI * (double)k / (double)n;
            f->tw_re[k] = (REAL)cos(a);
            f->tw_im[k] = (REAL)sin(a);
        }
        return 0;
    }
    int h = n / 2 > 0 ? n / 2 : 1;
    f->tw_re = (REAL *)malloc(sizeof(REAL) * (size_t)h);
    f->tw_im = (REAL *)malloc(sizeof(REAL) * (size_t)h);
    f->rev = (int *)malloc(sizeof(int) * (size_t)n);
    if (!f->tw_re || !f->tw_im || !f->rev)
        return -1;
    for (int k = 0; k < n / 2; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        f->tw_re[k] = (REAL)cos(a);
        f->tw_im[k] = (REAL)sin(a);
    }
    int bits = 0;
    while ((1 << bits) < n)
        ++bits;
    for (int i = 0; i < n; ++i) {
        int r = 0;
        for (int b = 0; b < bits; ++b)
            if (i & (1 << b))
                r |= 1 << (bits - 1 - b);
        f->rev[i] = r;
    }
    return 0;
}

static void SFX(fft_free)(SFX(ora_fft) * f)
{
    free(f->tw_re);
    free(f->tw_im);
    free(f->rev);
    free(f->ftw);
    free(f->fbuf);
}

/* Radix-4 decimation-in-frequency Stockham autosort FFT on planar (re, im) arrays; a trailing radix-2 stage when
 * log2 n is odd.  Stage with sub-length m (n, n/4, ...) and stride s = n/m: butterfly i = q + s p (p < m/4, q < s)
 * reads x[i + k n/4], k < 4 -- contiguous in i at every stage -- and writes y[q + s (4 p + j)]. */
static int SFX(fft_fast_plan)(SFX(ora_fft) * f)
{
    const int n = f->n;
    if (n < 4)
        return -1;
    int st = 0;
    for (int m = n; m >= 4; m /= 4)
        ++st;
    f->fast_stages = st;
    f->ftw = (REAL *)malloc(sizeof(REAL) * (size_t)st * 6 * (size_t)(n / 4));
    f->fbuf = (REAL *)malloc(sizeof(REAL) * 4 * (size_t)n);
    if (!f->ftw || !f->fbuf)
        return -1;
    int t = 0;
    for (int m = n, s = 1; m >= 4; m /= 4, s *= 4, ++t) {
        REAL *w = f->ftw + (size_t)t * 6 * (size_t)(n / 4);
        for (int i = 0; i < n / 4; ++i) {
            const int p = i / s;
            for (int k = 1; k <= 3; ++k) {
                const double a = -2.0 * M_PI * (double)(k * p) / (double)m;
                w[(2 * (k - 1)) * (n / 4) + i] = (REAL)cos(a);
                w[(2 * (k - 1) + 1) * (n / 4) + i] = (REAL)sin(a);
            }
        }
    }
    f->fast = 1;
    return 0;
}

static void SFX(fft_fast_forward)(const SFX(ora_fft) * f, REAL *c /* interleaved, in place */)
{
    const int n = f->n, q4 = n / 4;
    REAL *restrict xr = f->fbuf, *restrict xi = f->fbuf + n, *restrict yr = f->fbuf + 2 * n, *restrict yi = f->fbuf + 3 * n;
    for (int i = 0; i < n; ++i) {
        xr[i] = c[2 * i];
        xi[i] = c[2 * i + 1];
    }
    int t = 0, m = n, s = 1;
    for (; m >= 4; m /= 4, s *= 4, ++t) {
        const REAL *restrict w = f->ftw + (size_t)t * 6 * (size_t)q4;
        const REAL *restrict w1r = w, *restrict w1i = w + q4, *restrict w2r = w + 2 * q4, *restrict w2i = w + 3 * q4,
                   *restrict w3r = w + 4 * q4, *restrict w3i = w + 5 * q4;
        for (int p = 0; p < m / 4; ++p) {
            REAL *restrict o0r = yr + s * (4 * p), *restrict o0i = yi + s * (4 * p);
            const int i0 = s * p;
            for (int q = 0; q < s; ++q) {
                const int i = i0 + q;
                const REAL ar = xr[i], ai = xi[i], br = xr[i + q4], bi = xi[i + q4];
                const REAL cr = xr[i + 2 * q4], ci = xi[i + 2 * q4], dr = xr[i + 3 * q4], di = xi[i + 3 * q4];
                const REAL apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
                const REAL bpdr = br + dr, bpdi = bi + di;
                const REAL jr = -(bi - di), ji = br - dr; /* j (b - d) */
                const REAL t1r = amcr - jr, t1i = amci - ji, t2r = apcr - bpdr, t2i = apci - bpdi;
                const REAL t3r = amcr + jr, t3i = amci + ji;
                o0r[q] = apcr + bpdr;
                o0i[q] = apci + bpdi;
                o0r[q + s] = t1r * w1r[i] - t1i * w1i[i];
                o0i[q + s] = t1r * w1i[i] + t1i * w1r[i];
                o0r[q + 2 * s] = t2r * w2r[i] - t2i * w2i[i];
                o0i[q + 2 * s] = t2r * w2i[i] + t2i * w2r[i];
                o0r[q + 3 * s] = t3r * w3r[i] - t3i * w3i[i];
                o0i[q + 3 * s] = t3r * w3i[i] + t3i * w3r[i];
            }
        }
        REAL *tr = xr, *ti = xi;
        xr = yr;
        xi = yi;
        yr = tr;
        yi = ti;
    }
    if (m == 2) { /* last radix-2 stage: stride s = n/2 */
        for (int q = 0; q < s; ++q) {
            const REAL ar = xr[q], ai = xi[q], br = xr[q + s], bi = xi[q + s];
            yr[q] = ar + br;
            yi[q] = ai + bi;
            yr[q + s] = ar - br;
            yi[q + s] = ai - bi;
        }
        xr = yr;
        xi = yi;
    }
    for (int i = 0; i < n; ++i) {
        c[2 * i] = xr[i];
        c[2 * i + 1] = xi[i];
    }
}

/* in-place on interleaved (re, im) pairs */
static void SFX(fft_forward)(const SFX(ora_fft) * f, REAL *c)
{
    const int n = f->n;
    if (!f->rev) { /* the DFT by definition (sizes that are not powers of two) */
        REAL *y = f->fbuf;
        for (int k = 0; k < n; ++k) {
            REAL sr = 0, si = 0;
            int idx = 0; /* (j k) mod n, incrementally */
            for (int j = 0; j < n; ++j) {
                const REAL wr = f->tw_re[idx], wi = f->tw_im[idx];
                sr += c[2 * j] * wr - c[2 * j + 1] * wi;
                si += c[2 * j] * wi + c[2 * j + 1] * wr;
                idx += k;
                if (idx >= n)
                    idx -= n;
            }
            y[2 * k] = sr;
            y[2 * k + 1] = si;
        }
        memcpy(c, y, sizeof(REAL) * 2 * (size_t)n);
        return;
    }
    for (int i = 0; i < n; ++i) {
        int r = f->rev[i];
        if (r > i) {
            REAL tr = c[2 * i], ti = c[2 * i + 1];
            c[2 * i] = c[2 * r];
            c[2 * i + 1] = c[2 * r + 1];
            c[2 * r] = tr;
            c[2 * r + 1] = ti;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1, step = n / len;
        for (int b = 0; b < n; b += len) {
            for (int k = 0; k < half; ++k) {
                const REAL wr = f->tw_re[k * step], wi = f->tw_im[k * step];
                REAL *p = c + 2 * (b + k), *q = c + 2 * (b + k + half);
                const REAL xr = q[0] * wr - q[1] * wi;
                const REAL xi = q[0] * wi + q[1] * wr;
                q[0] = p[0] - xr;
                q[1] = p[1] - xi;
                p[0] = p[0] + xr;
                p[1] = p[1] + xi;
            }
        }
    }
}

/* ---- Detrend (src/psd.rs:59-114) ----------------------------------------- */
/* kind: 0 None, 1 Midpoint, 2 Span, 3 Mean (4 Linear: unimplemented!()) */

static int SFX(detrend_apply)(int kind, const REAL *x, const SFX(ora_window_t) * win,
                              REAL *c /* 2n interleaved */)
{
    const int n = win->n;
    switch (kind) {
    case 0: /* src/psd.rs:81-86 */
        for (int i = 0; i < n; ++i) {
            c[2 * i] = x[i] * win->win[i];
            c[2 * i + 1] = 0;
        }
        break;
    case 1: { /* src/psd.rs:87-93 */
        const REAL offset = x[n / 2];
        for (int i = 0; i < n; ++i) {
            c[2 * i] = (x[i] - offset) * win->win[i];
            c[2 * i + 1] = 0;
        }
        break;
    }
    case 2: { /* src/psd.rs:94-102: offset is a sequentially accumulated ramp */
        REAL offset = x[0];
        const REAL slope = (x[n - 1] - x[0]) / (REAL)(n - 1);
        for (int i = 0; i < n; ++i) {
            c[2 * i] = (x[i] - offset) * win->win[i];
            c[2 * i + 1] = 0;
            offset += slope;
        }
        break;
    }
    case 3: { /* src/psd.rs:103-109: sequential sum */
        REAL sum = 0;
        for (int i = 0; i < n; ++i)
            sum += x[i];
        const REAL offset = sum / (REAL)n;
        for (int i = 0; i < n; ++i) {
            c[2 * i] = (x[i] - offset) * win->win[i];
            c[2 * i + 1] = 0;
        }
        break;
    }
    default: /* src/psd.rs:110 unimplemented!() */
        return -1;
    }
    return 0;
}

/* ---- Half-band decimate-by-2 stage (idsp hbf: HbfDec = even delay + SymFir)
 *
 * y[j] = xe[j-(M-1)] + sum_{i<M} taps[i]*(xo[j-(2M-1)+i] + xo[j-i]),
 * xe[m]=x[2m], xo[m]=x[2m+1], zero initial state.  The sum runs i=0..M-1 in
 * order as (old+new)*tap, starting from 0, then even+odd (idsp SymFir::get /
 * HbfDec::process_block arithmetic order). */

typedef struct {
    int m;
    REAL taps[16];
    REAL even[16]; /* last M-1 even samples, oldest first */
    REAL odd[32];  /* last 2M-1 odd samples, oldest first */
} SFX(ora_hbf2);

static void SFX(hbf2_init)(SFX(ora_hbf2) * h, int m, const double *taps)
{
    memset(h, 0, sizeof(*h));
    h->m = m;
    for (int i = 0; i < m; ++i)
        h->taps[i] = (REAL)(float)taps[i]; /* idsp's HBF_TAPS are f32 constants: round, then widen */
}

/* Block form: x holds 2k samples, y receives k; e/o are scratch of at least
 * k + 2M entries.  Same arithmetic, per output, as the formula above. */
static void SFX(hbf2_block)(SFX(ora_hbf2) * h, const REAL *x, int k, REAL *y, REAL *e, REAL *o)
{
    const int m = h->m;
    memcpy(e, h->even, sizeof(REAL) * (size_t)(m - 1));
    memcpy(o, h->odd, sizeof(REAL) * (size_t)(2 * m - 1));
    for (int j = 0; j < k; ++j) {
        e[m - 1 + j] = x[2 * j];
        o[2 * m - 1 + j] = x[2 * j + 1];
    }
    /* per output: acc = 0; acc += (old + new) * tap for i = 0..M-1 in order; y = even + acc -- written with the tap
     * loop outside so that the compiler vectorises over the outputs (the same operations in the same order for
     * every output, hence the same bits) */
    for (int j = 0; j < k; ++j)
        y[j] = 0;
    for (int i = 0; i < m; ++i) {
        const REAL t = h->taps[i];
        const REAL *restrict oa = o + i, *restrict ob = o + 2 * m - 1 - i;
        for (int j = 0; j < k; ++j)
            y[j] += (oa[j] + ob[j]) * t;
    }
    for (int j = 0; j < k; ++j)
        y[j] = e[j] + y[j];
    memcpy(h->even, e + k, sizeof(REAL) * (size_t)(m - 1));
    memcpy(h->odd, o + k, sizeof(REAL) * (size_t)(2 * m - 1));
}

/* HbfDec8 = the last three /2 stages of HBF_DEC_CASCADE (src/psd.rs:248-253):
 * 3-tap stage at the input rate, then 6-tap, then 15-tap. */
typedef struct {
    SFX(ora_hbf2) s2, s1, s0;
    REAL *t0, *t1, *e, *o; /* scratch sized for blocks of up to cap inputs */
    int cap;
} SFX(ora_hbf8);

static int SFX(hbf8_init)(SFX(ora_hbf8) * h, int cap)
{
    SFX(hbf2_init)(&h->s2, ORA_HBF_M2, ORA_HBF_TAPS2);
    SFX(hbf2_init)(&h->s1, ORA_HBF_M1, ORA_HBF_TAPS1);
    SFX(hbf2_init)(&h->s0, ORA_HBF_M0, ORA_HBF_TAPS0);
    h->cap = cap;
    h->t0 = (REAL *)malloc(sizeof(REAL) * (size_t)(cap / 2 + 1));
    h->t1 = (REAL *)malloc(sizeof(REAL) * (size_t)(cap / 4 + 1));
    h->e = (REAL *)malloc(sizeof(REAL) * (size_t)(cap / 2 + 64));
    h->o = (REAL *)malloc(sizeof(REAL) * (size_t)(cap / 2 + 64));
    return (h->t0 && h->t1 && h->e && h->o) ? 0 : -1;
}

static void SFX(hbf8_free)(SFX(ora_hbf8) * h)
{
    free(h->t0);
    free(h->t1);
    free(h->e);
    free(h->o);
}

/* nb chunks of 8 inputs -> nb outputs (src/psd.rs:246-253) */
static void SFX(hbf8_block)(SFX(ora_hbf8) * h, const REAL *x, int nb, REAL *y)
{
    while (nb > 0) {
        int b = nb < h->cap / 8 ? nb : h->cap / 8;
        SFX(hbf2_block)(&h->s2, x, 4 * b, h->t0, h->e, h->o);
        SFX(hbf2_block)(&h->s1, h->t0, 2 * b, h->t1, h->e, h->o);
        SFX(hbf2_block)(&h->s0, h->t1, b, y, h->e, h->o);
        x += 8 * b;
        y += b;
        nb -= b;
    }
}

/* ---- Psd<N> (src/psd.rs:122-288) ----------------------------------------- */

typedef struct {
    int n;
    SFX(ora_hbf8) hbf;      /* :124 */
    REAL *buf;              /* :125 */
    int idx;                /* :126 */
    REAL *spectrum;         /* :127 (n entries, first n/2+1 used) */
    uint32_t count;         /* :128 */
    int drain;              /* :129 */
    const SFX(ora_fft) * fft;
    const SFX(ora_window_t) * win;
    int detrend;            /* :132 */
    uint32_t avg;           /* :133 */
    REAL *c;                /* scratch complex frame */
    int wrote_past_n;       /* a write went past y[N): the cascade's [f32; N] would have panicked */
} SFX(ora_psd);

static int SFX(psd_init)(SFX(ora_psd) * s, const SFX(ora_fft) * fft, const SFX(ora_window_t) * win)
{
    const int n = win->n;
    if (n < 2 || fft->n != n) /* :138-139 */
        return -1;
    s->n = n;
    if (SFX(hbf8_init)(&s->hbf, n < 8 ? 8 : n))
        return -1;
    s->buf = (REAL *)calloc((size_t)n, sizeof(REAL));
    s->spectrum = (REAL *)calloc((size_t)n, sizeof(REAL));
    s->c = (REAL *)calloc((size_t)n * 2, sizeof(REAL));
    if (!s->buf || !s->spectrum || !s->c)
        return -1;
    s->idx = 0;
    s->count = 0;
    s->wrote_past_n = 0;
    s->fft = fft;
    s->win = win;
    s->detrend = 0;
    s->drain = ora_hbf_dec_response_length(3); /* :149 */
    s->avg = UINT32_MAX;                       /* :150 */
    return 0;
}

static void SFX(psd_free)(SFX(ora_psd) * s)
{
    SFX(hbf8_free)(&s->hbf);
    free(s->buf);
    free(s->spectrum);
    free(s->c);
}

/* PsdStage::process (src/psd.rs:196-269).  Returns the number of items written
 * to y, or -1 on a contract violation (the reference panics). */
static long SFX(psd_process)(SFX(ora_psd) * s, const REAL *x, size_t xlen, REAL *y)
{
    const int n = s->n;
    const int ov = s->win->overlap;
    size_t nout = 0;
    while (xlen > 0) { /* :199 */
        /* load :201-208 */
        size_t take = xlen < (size_t)(n - s->idx) ? xlen : (size_t)(n - s->idx);
        memcpy(s->buf + s->idx, x, sizeof(REAL) * take);
        x += take;
        xlen -= take;
        s->idx += (int)take;
        if (s->idx < n)
            break;

        /* detrend and window :211, fft :213 */
        if (SFX(detrend_apply)(s->detrend, s->buf, s->win, s->c))
            return -1;
        if (s->fft->fast)
            SFX(fft_fast_forward)(s->fft, s->c);
        else
            SFX(fft_forward)(s->fft, s->c);

        const int is_first = s->count == 0; /* :215 */

        /* EWMA :218-225 */
        REAL g;
        if (s->count > s->avg) {
            /* `avg as f32 / count as f32` (:220): the factor is DEFINED in f32 -- its rounding (3e-8) is part of
             * the algorithm, and 1/(1 - g) turns it into 3e-5 of a long exponential average -- so the f64
             * instantiation takes the f32 quotient too and only widens it */
            g = (REAL)((float)s->avg / (float)s->count);
            s->count = s->avg;
        } else {
            g = (REAL)1.0;
        }
        s->count += 1;

        /* power + accumulate :228-233 (norm_sqr = re*re + im*im) */
        for (int k = 0; k <= n / 2; ++k) {
            const REAL re = s->c[2 * k], im = s->c[2 * k + 1];
            s->spectrum[k] = g * s->spectrum[k] + (re * re + im * im);
        }

        int start; /* :235-243 */
        if (is_first) {
            start = 0;
        } else {
            memmove(s->buf, s->buf + (n - ov), sizeof(REAL) * (size_t)ov);
            start = ov;
        }

        /* decimate :246-253 */
        if ((n - start) % 8 != 0)
            return -1; /* assert!(xr.is_empty()) :247 */
        const int nb = (n - start) / 8;
        if (nout + (size_t)nb > (size_t)n)
            s->wrote_past_n = 1; /* y[n..][..xb.len()] with y: [f32; N] would panic (:253, :457) */
        SFX(hbf8_block)(&s->hbf, s->buf + start, nb, y + nout);
        /* drain :255-260 */
        int skip = s->drain < nb ? s->drain : nb;
        if (skip > 0) {
            s->drain -= skip;
            memmove(y + nout, y + nout + skip, sizeof(REAL) * (size_t)(nb - skip));
        }
        nout += (size_t)(nb - skip);

        if (is_first) /* :262-265 */
            memmove(s->buf, s->buf + (n - ov), sizeof(REAL) * (size_t)ov);
        s->idx = ov; /* :266 */
    }
    return (long)nout;
}

/* gain (src/psd.rs:279-283): u32 multiply (wrapping in release), then f32 */
static REAL SFX(psd_gain)(const SFX(ora_psd) * s)
{
    uint32_t m = (uint32_t)s->n / 2u * s->count;
    return (REAL)m * s->win->nenbw * s->win->power;
}

/* ---- PsdCascade<N> (src/psd.rs:399-544) ---------------------------------- */

#define ORA_MAX_STAGES 24

typedef struct SFX(ora_cascade)
{
    int n;
    SFX(ora_fft) fft;
    SFX(ora_window_t) win;
    int detrend;
    uint32_t avg_limit, avg_count;
    int n_stages;
    SFX(ora_psd) stages[ORA_MAX_STAGES];
    REAL *a0, *a1, *xin; /* ping-pong + converted input chunk */
    /* Set when a stage wrote past N items of its output buffer.  The reference's ping-pong buffers
     * are [f32; N] (src/psd.rs:457-458) although PsdStage::process asks for x.len()/8 + N/8
     * (SURVEY.md row A4): when a call completes 16 segments INCLUDING the stream's first one (some
     * samples were buffered by an earlier short call, then a full 8N chunk arrives) stage 0 emits
     * N + N/16 - 35 items and `&mut y[n..][..xb.len()]` (src/psd.rs:253) panics.  The oracle sizes
     * its buffers as the contract asks and keeps going, so that chunk invariance can be tested on
     * such feeds too; this flag says the reference would have panicked. */
    int ref_would_panic;
} SFX(ora_cascade);

static uint32_t SFX(stage_avg)(const SFX(ora_cascade) * c, int i)
{
    /* (self.avg.count >> (DEPTH * i)).min(self.avg.limit)  :434,:449
     * (shift >= 32 is a latent overflow in the reference; treated as 0) */
    uint32_t v = (3 * i >= 32) ? 0u : (c->avg_count >> (3 * i));
    return v < c->avg_limit ? v : c->avg_limit;
}

SFX(ora_cascade) * SFX(ora_cascade_new)(int n, int window_kind)
{
    SFX(ora_cascade) *c = (SFX(ora_cascade) *)calloc(1, sizeof(*c));
    if (!c)
        return NULL;
    c->n = n;
    if (SFX(fft_init)(&c->fft, n) || SFX(window_init)(&c->win, n, window_kind)) {
        free(c);
        return NULL;
    }
    c->detrend = 0;             /* :418 */
    c->avg_limit = UINT32_MAX;  /* :369-375 */
    c->avg_count = UINT32_MAX;
    c->a0 = (REAL *)calloc((size_t)n + (size_t)n / 8 + 8, sizeof(REAL)); /* x.len()/8 + N/8 */
    c->a1 = (REAL *)calloc((size_t)n + (size_t)n / 8 + 8, sizeof(REAL));
    c->xin = (REAL *)calloc((size_t)n * 8, sizeof(REAL));
    return c;
}

/* PsdCascade with a caller-built window: the reference's Default builds Hann (src/psd.rs:419); a cascade over
 * another Window<N> is what `stages: Vec<Psd<N>>` + `win: Arc<Window<N>>` (:401-404) hold when constructed by hand */
SFX(ora_cascade) * SFX(ora_cascade_new_window)(int n, const float *win, float power, float nenbw, int overlap)
{
    SFX(ora_cascade) *c = (SFX(ora_cascade) *)calloc(1, sizeof(*c));
    if (!c)
        return NULL;
    c->n = n;
    if (SFX(fft_init)(&c->fft, n) || SFX(window_init_table)(&c->win, n, win, power, nenbw, overlap)) {
        free(c);
        return NULL;
    }
    c->detrend = 0;
    c->avg_limit = UINT32_MAX;
    c->avg_count = UINT32_MAX;
    c->a0 = (REAL *)calloc((size_t)n + (size_t)n / 8 + 8, sizeof(REAL));
    c->a1 = (REAL *)calloc((size_t)n + (size_t)n / 8 + 8, sizeof(REAL));
    c->xin = (REAL *)calloc((size_t)n * 8, sizeof(REAL));
    return c;
}

void SFX(ora_cascade_free)(SFX(ora_cascade) * c)
{
    if (!c)
        return;
    for (int i = 0; i < c->n_stages; ++i)
        SFX(psd_free)(&c->stages[i]);
    SFX(fft_free)(&c->fft);
    free(c->win.win);
    free(c->a0);
    free(c->a1);
    free(c->xin);
    free(c);
}

/* cpu_baseline only: switch this cascade's FFT to the vectorisable radix-4 plan (same DFT, other rounding) */
int SFX(ora_cascade_set_fast_fft)(SFX(ora_cascade) * c)
{
    if (!c->fft.rev)
        return -1; /* powers of two only */
    return c->fft.fast ? 0 : SFX(fft_fast_plan)(&c->fft);
}

void SFX(ora_cascade_set_avg)(SFX(ora_cascade) * c, uint32_t limit, uint32_t count)
{ /* :431-436 */
    c->avg_limit = limit;
    c->avg_count = count;
    for (int i = 0; i < c->n_stages; ++i)
        c->stages[i].avg = SFX(stage_avg)(c, i);
}

int SFX(ora_cascade_set_detrend)(SFX(ora_cascade) * c, int kind)
{ /* :438-443 */
    if (kind < 0 || kind > 3)
        return -1;
    c->detrend = kind;
    for (int i = 0; i < c->n_stages; ++i)
        c->stages[i].detrend = kind;
    return 0;
}

static SFX(ora_psd) * SFX(get_or_add)(SFX(ora_cascade) * c, int i)
{ /* :445-453 */
    while (i >= c->n_stages) {
        if (c->n_stages >= ORA_MAX_STAGES)
            return NULL;
        SFX(ora_psd) *s = &c->stages[c->n_stages];
        if (SFX(psd_init)(s, &c->fft, &c->win))
            return NULL;
        s->detrend = c->detrend;
        s->avg = SFX(stage_avg)(c, c->n_stages);
        c->n_stages++;
    }
    return &c->stages[i];
}

/* PsdCascade::process (src/psd.rs:456-468); input samples are always f32 */
int SFX(ora_cascade_process)(SFX(ora_cascade) * c, const float *x, size_t len)
{
    const size_t chunk = (size_t)c->n << 3; /* N << DEPTH :459 */
    while (len > 0) {
        size_t m = len < chunk ? len : chunk;
        for (size_t i = 0; i < m; ++i)
            c->xin[i] = (REAL)x[i];
        const REAL *xp = c->xin;
        size_t xl = m;
        REAL *y = c->a0, *z = c->a1;
        int i = 0;
        while (xl > 0) { /* :461 */
            SFX(ora_psd) *s = SFX(get_or_add)(c, i);
            if (!s)
                return -1;
            long nn = SFX(psd_process)(s, xp, xl, y);
            if (nn < 0)
                return -1;
            if (s->wrote_past_n)
                c->ref_would_panic = 1;
            REAL *t = z; /* swap :463 */
            z = y;
            y = t;
            xp = z; /* :464 */
            xl = (size_t)nn;
            i += 1;
        }
        x += m;
        len -= m;
    }
    return 0;
}

int SFX(ora_cascade_num_stages)(const SFX(ora_cascade) * c) { return c->n_stages; }

/* 1 if some call so far would have made the reference panic at src/psd.rs:253 (see ref_would_panic) */
int SFX(ora_cascade_ref_would_panic)(const SFX(ora_cascade) * c) { return c->ref_would_panic; }

typedef struct {
    uint32_t count;
    uint32_t avg;
    uint64_t pending;   /* buf().len() :285-287 */
    uint64_t processed; /* :511-512 */
} SFX(ora_stage_info);

int SFX(ora_cascade_stage_info)(const SFX(ora_cascade) * c, int stage, uint32_t *count,
                                uint32_t *avg, uint64_t *pending, uint64_t *processed)
{
    if (stage < 0 || stage >= c->n_stages)
        return -1;
    const SFX(ora_psd) *s = &c->stages[stage];
    *count = s->count;
    *avg = s->avg;
    *pending = (uint64_t)s->idx;
    uint32_t cm1 = s->count ? s->count - 1 : 0; /* saturating_sub(1) */
    *processed = (uint64_t)c->n * s->count - (uint64_t)s->win->overlap * cm1;
    return 0;
}

int SFX(ora_cascade_stage_spectrum)(const SFX(ora_cascade) * c, int stage, REAL *out)
{
    if (stage < 0 || stage >= c->n_stages)
        return -1;
    memcpy(out, c->stages[stage].spectrum, sizeof(REAL) * (size_t)(c->n / 2 + 1));
    return 0;
}

int SFX(ora_cascade_stage_buf)(const SFX(ora_cascade) * c, int stage, REAL *out)
{
    if (stage < 0 || stage >= c->n_stages)
        return -1;
    memcpy(out, c->stages[stage].buf, sizeof(REAL) * (size_t)c->stages[stage].idx);
    return 0;
}

REAL SFX(ora_cascade_stage_gain)(const SFX(ora_cascade) * c, int stage)
{
    if (stage < 0 || stage >= c->n_stages)
        return 0;
    return SFX(psd_gain)(&c->stages[stage]);
}

/* PsdCascade::psd (src/psd.rs:479-543).  breaks: n_stages records, lowest
 * rate first.  Returns the merged length; psd_out needs n_stages*(n/2+1). */
long SFX(ora_cascade_psd)(const SFX(ora_cascade) * c, int keep_overlap, uint32_t min_count,
                          int keep_transition_band, REAL *psd_out, ora_break *breaks)
{
    const int n = c->n;
    size_t plen = 0;
    int nb = 0;
    uint64_t decimation = 1ull << (3 * c->n_stages); /* :482 */
    size_t end = 0;
    for (int si = c->n_stages - 1; si >= 0; --si) { /* .rev() :484 */
        const SFX(ora_psd) *s = &c->stages[si];
        decimation >>= 3;
        size_t start = keep_overlap ? 0 : ((end + 7) >> 3); /* :490-495 */
        end = (decimation > 1 && !keep_transition_band) ? (size_t)(2 * n / 5)
                                                        : (size_t)(n / 2 + 1); /* :496-501 */
        int include = s->count >= min_count; /* :502 */
        ora_break *b = &breaks[nb++];
        b->start = plen;
        b->include = include;
        b->count = s->count;
        b->avg = s->avg;
        b->bins_start = start;
        b->bins_end = end;
        b->fft_size = (uint64_t)n;
        b->decimation = decimation;
        uint32_t cm1 = s->count ? s->count - 1 : 0;
        b->processed = (uint64_t)n * s->count - (uint64_t)s->win->overlap * cm1;
        b->pending = (uint64_t)s->idx;
        if (include) { /* :515-517 */
            const REAL g = (REAL)1.0 / (SFX(psd_gain)(s) * (REAL)decimation);
            for (size_t k = start; k < end; ++k)
                psd_out[plen++] = s->spectrum[k] * g;
        } else {
            end = start; /* :518-520 */
        }
    }
    return (long)plen;
}

/* ---- single-stage access (src/psd.rs:615-632 test shape) ----------------- */

typedef struct SFX(ora_stage)
{
    SFX(ora_fft) fft;
    SFX(ora_window_t) win;
    SFX(ora_psd) psd;
} SFX(ora_stage);

SFX(ora_stage) * SFX(ora_stage_new)(int n, int window_kind)
{
    SFX(ora_stage) *s = (SFX(ora_stage) *)calloc(1, sizeof(*s));
    if (!s)
        return NULL;
    if (SFX(fft_init)(&s->fft, n) || SFX(window_init)(&s->win, n, window_kind) ||
        SFX(psd_init)(&s->psd, &s->fft, &s->win)) {
        free(s);
        return NULL;
    }
    return s;
}

/* Psd::new(fft, win) with a caller-built Window<N> (src/psd.rs:137-152) */
SFX(ora_stage) * SFX(ora_stage_new_window)(int n, const float *win, float power, float nenbw, int overlap)
{
    SFX(ora_stage) *s = (SFX(ora_stage) *)calloc(1, sizeof(*s));
    if (!s)
        return NULL;
    if (SFX(fft_init)(&s->fft, n) || SFX(window_init_table)(&s->win, n, win, power, nenbw, overlap) ||
        SFX(psd_init)(&s->psd, &s->fft, &s->win)) {
        free(s);
        return NULL;
    }
    return s;
}

void SFX(ora_stage_free)(SFX(ora_stage) * s)
{
    if (!s)
        return;
    SFX(psd_free)(&s->psd);
    SFX(fft_free)(&s->fft);
    free(s->win.win);
    free(s);
}

void SFX(ora_stage_set)(SFX(ora_stage) * s, int detrend, uint32_t avg)
{
    s->psd.detrend = detrend;
    s->psd.avg = avg;
}

/* y must hold xlen/8 + n/8 items (src/psd.rs:196 contract) */
long SFX(ora_stage_process)(SFX(ora_stage) * s, const float *x, size_t xlen, REAL *y)
{
    REAL *xr = (REAL *)malloc(sizeof(REAL) * (xlen ? xlen : 1));
    if (!xr)
        return -1;
    for (size_t i = 0; i < xlen; ++i)
        xr[i] = (REAL)x[i];
    long r = SFX(psd_process)(&s->psd, xr, xlen, y);
    free(xr);
    return r;
}

void SFX(ora_stage_spectrum)(const SFX(ora_stage) * s, REAL *out)
{
    memcpy(out, s->psd.spectrum, sizeof(REAL) * (size_t)(s->psd.n / 2 + 1));
}
REAL SFX(ora_stage_gain)(const SFX(ora_stage) * s) { return SFX(psd_gain)(&s->psd); }
uint32_t SFX(ora_stage_count)(const SFX(ora_stage) * s) { return s->psd.count; }
int SFX(ora_stage_pending)(const SFX(ora_stage) * s) { return s->psd.idx; }

/* ---- building blocks exposed for unit tests ------------------------------ */

int SFX(ora_window)(int n, int kind, REAL *win_out, REAL *power, REAL *nenbw, int *overlap)
{
    SFX(ora_window_t) w;
    if (SFX(window_init)(&w, n, kind))
        return -1;
    memcpy(win_out, w.win, sizeof(REAL) * (size_t)n);
    *power = w.power;
    *nenbw = w.nenbw;
    *overlap = w.overlap;
    free(w.win);
    return 0;
}

int SFX(ora_fft_forward)(int n, REAL *c_interleaved)
{
    SFX(ora_fft) f;
    if (SFX(fft_init)(&f, n))
        return -1;
    SFX(fft_forward)(&f, c_interleaved);
    SFX(fft_free)(&f);
    return 0;
}

int SFX(ora_detrend_apply)(int n, int window_kind, int detrend, const REAL *x, REAL *c)
{
    SFX(ora_window_t) w;
    if (SFX(window_init)(&w, n, window_kind))
        return -1;
    int r = SFX(detrend_apply)(detrend, x, &w, c);
    free(w.win);
    return r;
}

/* zero-state /8 decimation of a whole array: nout = len/8 */
long SFX(ora_hbf_dec8)(const REAL *x, size_t len, REAL *y)
{
    SFX(ora_hbf8) h;
    if (SFX(hbf8_init)(&h, 4096))
        return -1;
    size_t nb = len / 8, done = 0;
    while (done < nb) {
        int b = (nb - done) < 512 ? (int)(nb - done) : 512;
        SFX(hbf8_block)(&h, x + 8 * done, b, y + done);
        done += (size_t)b;
    }
    SFX(hbf8_free)(&h);
    return (long)nb;
}
