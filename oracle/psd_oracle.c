/*
 * TEST INFRASTRUCTURE ONLY (oracle).  Not part of the shipped library.
 *
 * CPU restatement of the cascaded PSD hot path of quartiq/stabilizer-stream
 * (src/psd.rs: Window, Detrend, Psd, PsdStage, Break, MergeOpts, AvgOpts,
 * PsdCascade) plus src/var.rs Var::eval and the payload decoders
 * (src/de/frame.rs, src/de/data.rs:11-212: AdcDac, Fls, ThermostatEem, Mpll).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The shipped library (stabilizer-stream_amd/csrc) never does.
 *
 * PARITY STATUS
 *   - window, detrend, segment loop, EWMA, overlap, drain, gain, stitch,
 *     frequencies: restated line by line from src/psd.rs (cited inline) and
 *     pinned by the reference's own known answers (tests/golden/reference_*.json:
 *     N=4 Hann [16/3,4/3,0] src/psd.rs:588-595; Hann constants :49-54; length
 *     relation :622; white-noise bounds :623-643; Var 0.13478442 src/var.rs:55-59).
 *   - FFT: rustfft 6.4.1 is not in the container; the forward unnormalised DFT
 *     is mathematically defined (src/psd.rs:213, :279-283) and checked against
 *     numpy.  Rounding differs at the 1e-7 level.
 *   - half-band decimator: idsp 0.20.0 is not in the container.  Taps and
 *     response length are restated from the published crate (hbf_taps_oracle.h);
 *     PARITY UNPINNED for exact decimator output samples, pinned only by the
 *     reference's statistical assertions.
 *   - payload decoders: AdcDac constants are the reference's (data.rs:28-35).  The
 *     reference holds NO test or fixture for Fls / ThermostatEem / Mpll: PARITY
 *     UNPINNED by reference-held vectors for those three -- pinned by hand-computed
 *     known answers and a second independent (numpy) restatement that must agree
 *     bit for bit (tests/test_payload_formats_oracle.py).
 *   - The Rust reference cannot be built here (no rustc/cargo; dependencies
 *     not vendored): there is no oracle/_ref.
 *
 * Two instantiations: *_f32 mirrors the reference's f32 arithmetic (compile
 * with -ffp-contract=off: rustc does not fuse a*b+c), *_f64 is the truth.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#include "hbf_taps_oracle.h"

/* Break (src/psd.rs:290-311) with bins: Range<usize> flattened */
typedef struct {
    uint64_t start;
    uint32_t include;
    uint32_t count;
    uint32_t avg;
    uint32_t _pad;
    uint64_t bins_start, bins_end;
    uint64_t fft_size;
    uint64_t decimation;
    uint64_t pending;
    uint64_t processed;
} ora_break;

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define SFX(name) CAT(name, _f32)
#define R_IS_F32 1
#include "psd_oracle_impl.h"
#undef REAL
#undef SFX
#undef R_IS_F32

#define REAL double
#define SFX(name) CAT(name, _f64)
#define R_IS_F32 0
#include "psd_oracle_impl.h"
#undef REAL
#undef SFX
#undef R_IS_F32

int ora_hbf_response_length(int depth) { return ora_hbf_dec_response_length(depth); }

/* Break::frequencies (src/psd.rs:315-327), rbw (:334-336) in f32 */
long ora_frequencies(const ora_break *b, int n, float *out)
{
    long len = 0;
    for (int i = 0; i < n; ++i) {
        if (!b[i].include)
            continue;
        const float rbw = 1.0f / (float)(b[i].fft_size * b[i].decimation);
        for (uint64_t f = b[i].bins_start; f < b[i].bins_end; ++f)
            out[len++] = (float)f * rbw;
    }
    return len;
}

/* Var::eval (src/var.rs:26-45), f32 like the reference.
 * defaults: x_exp=-2, sinx_exp=4, clip=f32::MAX, dc_cut=2 (src/var.rs:7-17) */
static float powi_f32(float x, int e)
{
    /* f32::powi: repeated multiplication (llvm.powi); negative -> reciprocal */
    int neg = e < 0;
    unsigned u = (unsigned)(neg ? -e : e);
    float r = 1.0f, b = x;
    while (u) {
        if (u & 1u)
            r *= b;
        b *= b;
        u >>= 1;
    }
    return neg ? 1.0f / r : r;
}

float ora_var_eval(int x_exp, int sinx_exp, float clip, size_t dc_cut, const float *phase_psd,
                   const float *frequencies, size_t n, float tau)
{
    float accu = 0.0f, a0 = 0.0f, f0 = 0.0f;
    const float pi = 3.14159265358979323846f;
    for (size_t i = dc_cut; i < n; ++i) {
        const float f = frequencies[i], sp = phase_psd[i];
        if (!(f <= clip / tau))
            break; /* take_while */
        const float sy = sp * f * f;
        const float pft = pi * (f * tau);
        const float hahd = powi_f32(sinf(pft), sinx_exp) * powi_f32(pft, x_exp);
        const float a = sy * hahd;
        accu = accu + (a + a0) * (f - f0);
        a0 = a;
        f0 = f;
    }
    return accu;
}

/* ---- Trace::plot + Trapezoidal (src/bin/psd.rs:98-116, :125-157), f32 like the reference.
 * Returns the number of plot points; *rms = sqrt(pi) (:156); plot_xy (may be NULL) gets
 * [log10(f) + log10(fs), integrate ? sqrt(p0.get()) : 10 (log10(p) - log10(fs))] per normal f. */
struct ora_trapezoidal {
    float x, y, i; /* :98-102, Default: zeros */
};
static float ora_trapezoidal_push(struct ora_trapezoidal *t, float x, float y)
{
    const float di = (y + t->y) * 0.5f * (x - t->x); /* :106 */
    t->x = x;                                        /* :107 */
    t->y = y;                                        /* :108 */
    t->i += di;                                      /* :109 */
    return di;
}
long ora_trace_plot(const float *psd, const float *frequencies, size_t n, float fs, int integrate,
                    float integral_start, float integral_end, float *rms, double *plot_xy)
{
    const float logfs = log10f(fs); /* :127 */
    struct ora_trapezoidal p0 = {0.0f, 0.0f, 0.0f};
    float pi = 0.0f;
    long np = 0;
    for (size_t k = 0; k < n; ++k) { /* psd.iter().zip(frequencies.iter()) :131-134 */
        const float p = psd[k], f = frequencies[k];
        const float dp = ora_trapezoidal_push(&p0, f, p); /* :136 */
        const float hz = fs * f;
        if (integral_start <= hz && hz <= integral_end) /* (start..=end).contains :137 */
            pi += dp;
        if (isnormal(f)) { /* :141 */
            if (plot_xy) {
                plot_xy[2 * np] = (double)(log10f(f) + logfs);                                          /* :143 */
                plot_xy[2 * np + 1] = (double)(integrate ? sqrtf(p0.i) : 10.0f * (log10f(p) - logfs)); /* :144-148 */
            }
            ++np;
        }
    }
    *rms = sqrtf(pi);
    return np;
}

/* ---- AdcDac frame decode (src/de/frame.rs:5-37,49-60; src/de/data.rs:11-82)
 * returns 0 ok; -1 InvalidHeader; -2 UnknownFormat; -3 PayloadSize;
 * -4 would panic in the reference (len<8, or len/64 != batches); -5 other format.
 * traces: 4 arrays (ADC0, ADC1, DAC0, DAC1) of 8*batches f32 each. */
int ora_adcdac_decode(const uint8_t *frame, size_t len, float *adc0, float *adc1, float *dac0,
                      float *dac1, uint32_t *seq, uint32_t *batches)
{
    if (len < 8)
        return -4; /* input[..HEADER_SIZE] panics (frame.rs:50) */
    if (frame[0] != 0x7b || frame[1] != 0x05)
        return -1; /* frame.rs:27-29 */
    const uint8_t fmt = frame[2];
    if (fmt < 1 || fmt > 4)
        return -2; /* frame.rs:30 */
    const uint32_t nb = frame[3];
    *seq = (uint32_t)frame[4] | ((uint32_t)frame[5] << 8) | ((uint32_t)frame[6] << 16) |
           ((uint32_t)frame[7] << 24);
    *batches = nb;
    if (fmt != 1)
        return -5;
    const size_t plen = len - 8;
    if (plen % 64 != 0)
        return -3; /* bytemuck::try_cast_slice (data.rs:23) */
    if (plen / 64 != nb)
        return -4; /* assert_eq!(data.len(), batches) (data.rs:24) */
    /* data.rs:31-35 */
    const float lsb = 4.096f * 2.5f / 32768.0f;
    float *tr[4] = {adc0, adc1, dac0, dac1};
    const uint8_t *p = frame + 8;
    for (uint32_t b = 0; b < nb; ++b) {
        for (int ch = 0; ch < 4; ++ch) {
            for (int i = 0; i < 8; ++i) {
                const uint8_t *q = p + ((size_t)b * 4 + (size_t)ch) * 16 + (size_t)i * 2;
                int16_t v = (int16_t)((uint16_t)q[0] | ((uint16_t)q[1] << 8));
                if (ch >= 2) /* i16.wrapping_add(i16::MIN) (data.rs:64,75) */
                    v = (int16_t)((uint16_t)v + 0x8000u);
                tr[ch][(size_t)b * 8 + (size_t)i] = (float)v * lsb;
            }
        }
    }
    return 0;
}

/* ---- the other payload formats (src/de/mod.rs:9-17; src/de/frame.rs:49-60; src/de/data.rs:84-212): Fls (id 2), ThermostatEem
 * (3), Mpll (4) -- ONE sample per batch and trace.  Any of the four formats:
 * returns 0 ok; -1 InvalidHeader; -2 UnknownFormat; -3 PayloadSize; -4 would panic in the reference (len < 8, or
 * payload / batch size != batches).  *fmt: the header's id; *ntraces 4, 4, 4, 3; tr[t] receives *nsamples f32 each
 * (8 * batches for AdcDac, batches otherwise; every tr[t] must hold (len - 8) / 8 + 8 floats). */
static int32_t ora_le_i32(const uint8_t *q)
{
    return (int32_t)((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24));
}
static float ora_le_f32(const uint8_t *q)
{
    const uint32_t u = (uint32_t)ora_le_i32(q);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
int ora_frame_decode(const uint8_t *frame, size_t len, float *tr0, float *tr1, float *tr2, float *tr3, uint32_t *fmt,
                     uint32_t *ntraces, uint32_t *nsamples, uint32_t *seq, uint32_t *batches)
{
    *fmt = *ntraces = *nsamples = *seq = *batches = 0;
    if (len < 8)
        return -4; /* input[..HEADER_SIZE] panics (frame.rs:50) */
    if (frame[0] != 0x7b || frame[1] != 0x05)
        return -1; /* frame.rs:27-29 */
    const uint8_t id = frame[2];
    if (id < 1 || id > 4)
        return -2; /* frame.rs:30, mod.rs:12-17 */
    *fmt = id;
    const uint32_t nb = frame[3];
    *seq = (uint32_t)ora_le_i32(frame + 4);
    *batches = nb;
    if (id == 1) {
        *ntraces = 4;
        const int st = ora_adcdac_decode(frame, len, tr0, tr1, tr2, tr3, seq, batches);
        if (st == 0)
            *nsamples = 8 * nb;
        return st;
    }
    /* bytes per batch: [[[u8;4];7];2] (data.rs:86), [[u8;4];16+4] (:144), [[u8;4];6] (:168) */
    const size_t bb = id == 2 ? 56 : id == 3 ? 80 : 24;
    const size_t plen = len - 8;
    if (plen % bb != 0)
        return -3; /* bytemuck::try_cast_slice (data.rs:91,149,173) */
    if (plen / bb != nb)
        return -4; /* assert_eq!(batches, data.len()) (data.rs:93,150,174) */
    *ntraces = id == 4 ? 3 : 4;
    *nsamples = nb;
    const uint8_t *p = frame + 8;
    for (uint32_t b = 0; b < nb; ++b, p += bb) {
        if (id == 2) { /* Fls::traces, data.rs:97-139 */
            const float re = (float)ora_le_i32(p), im = (float)ora_le_i32(p + 4);
            /* (re as f32).powi(2) + (im as f32).powi(2), .sqrt(), * (1.0 / (i32::MAX as f32)) :104-107 */
            tr0[b] = sqrtf(re * re + im * im) * (1.0f / (float)INT32_MAX);
            /* b[0][2..4] as one i64 (:114-119), * (TAU / (1i64 << 16) as f32) */
            const int64_t ph = (int64_t)((uint64_t)(uint32_t)ora_le_i32(p + 8) | ((uint64_t)(uint32_t)ora_le_i32(p + 12) << 32));
            tr1[b] = (float)ph * (6.28318530717958647692f / (float)(1ll << 16));
            tr2[b] = (float)ora_le_i32(p + 28) / (float)INT32_MAX; /* b[1][0] :127 */
            tr3[b] = (float)ora_le_i32(p + 32) / (float)INT32_MAX; /* b[1][1] :134 */
        } else if (id == 3) { /* ThermostatEem::traces, data.rs:154-163: words 0, 8, 13, 16 as f32 */
            tr0[b] = ora_le_f32(p);
            tr1[b] = ora_le_f32(p + 4 * 8);
            tr2[b] = ora_le_f32(p + 4 * 13);
            tr3[b] = ora_le_f32(p + 4 * 16);
        } else { /* Mpll::traces, data.rs:178-211 */
            const float two32 = (float)(1ull << 32);
            tr0[b] = (float)ora_le_i32(p + 16) * (6.28318530717958647692f / two32);      /* phase (rad) :184 */
            tr1[b] = (float)ora_le_i32(p + 20) * (1.0f / 1.28e-3f / two32);              /* frequency (kHz) :193 */
            const float x0 = (float)ora_le_i32(p), x1 = (float)ora_le_i32(p + 4);
            tr2[b] = sqrtf(x0 * x0 + x1 * x1) * (10.24f / 10.0f * 2.0f * 2.0f / two32); /* amplitude (V/G10) :202-206 */
        }
    }
    return 0;
}
