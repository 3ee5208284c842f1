/*
 * fft_oracle.c -- plain-C CPU restatement of the reference's algorithm for the hot path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file (oracle/liboracle.so).
 *
 * It follows the same pass algebra as oracle/pyfft_oracle.py (see that file's header for how the
 * oracle is pinned): a batched power-of-two c2c FFT executed as the chain of passes the reference
 * (fjarri-attic/pyfft 0.3.9) would launch -- citations are file:line in the reference tree:
 *
 *   radix lists     getGlobalRadixInfo   pyfft/kernel_helpers.py:67-122  (base 128 for multi-launch axes)
 *                   getRadixArray        pyfft/kernel_helpers.py:10-65   (table for LDS-resident axes)
 *   chain selection FFTPlan._fft1D       pyfft/plan.py:135-171
 *   one pass        globalKernel         pyfft/kernel.mako:805-1047:
 *        in  viewed as [outer][R][M][S], out as [outer][M][R][S]
 *        out[l][q][j] = w(R*M)^(l*q) * sum_r in[r][l][j] * w(R)^(r*q),  w(m) = exp(dir*2*pi*i/m)
 *   local kernel    localKernel          pyfft/kernel.mako:725-803 (same recursion, stride 1)
 *   scale rule      getScaleCoeffFunc    pyfft/kernel.py:23-37, applied by the last kernel (plan.py:125-128)
 *
 * Complex data are interleaved (re, im) pairs.  `prec` 0 = float32 arithmetic, 1 = float64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_PASSES 64

typedef struct { int R; int64_t M, S; } opass;

static int ilog2(int64_t n) { int r = 0; while (n > 1) { n >>= 1; ++r; } return r; }

/* kernel_helpers.py:67-122 (only the radix list matters for the arithmetic) */
static int global_radices(int64_t n, int* radix) {
    int64_t base = n < 128 ? n : 128, N = n;
    int cnt = 0;
    while (N > base) { N /= base; radix[cnt++] = (int)base; }
    radix[cnt++] = (int)N;
    return cnt;
}

/* kernel_helpers.py:49-65 */
static int local_radices(int64_t n, int* radix) {
    switch (n) {
        case 2: radix[0] = 2; return 1;
        case 4: radix[0] = 4; return 1;
        case 8: radix[0] = 8; return 1;
        case 16: radix[0] = 8; radix[1] = 2; return 2;
        case 32: radix[0] = 8; radix[1] = 4; return 2;
        case 64: radix[0] = 8; radix[1] = 8; return 2;
        case 128: radix[0] = 8; radix[1] = 4; radix[2] = 4; return 3;
        case 256: radix[0] = 4; radix[1] = 4; radix[2] = 4; radix[3] = 4; return 4;
        case 512: radix[0] = 8; radix[1] = 8; radix[2] = 8; return 3;
        case 1024: radix[0] = 16; radix[1] = 16; radix[2] = 4; return 3;
        case 2048: radix[0] = 8; radix[1] = 8; radix[2] = 8; radix[3] = 4; return 4;
    }
    return 0;
}

static int add_chain(opass* p, int np, const int* radix, int nr, int64_t n, int64_t radix_init) {
    int64_t S = radix_init, curr_n = n;
    for (int i = 0; i < nr; ++i) {
        p[np].R = radix[i];
        p[np].M = curr_n / radix[i];
        p[np].S = S;
        ++np;
        S *= radix[i];
        curr_n /= radix[i];
    }
    return np;
}

/* plan.py:111-171: X kernels, then the Y chain, then the Z chain */
int oracle_plan(int64_t x, int64_t y, int64_t z, int prec, int* R, int64_t* M, int64_t* S) {
    opass p[MAX_PASSES];
    int radix[32], np = 0, nr;
    int64_t max_smem = prec ? 1024 : 2048; /* plan.py:32,46 */
    if (x > max_smem) { nr = global_radices(x, radix); np = add_chain(p, np, radix, nr, x, 1); }
    else if (x > 1) { nr = local_radices(x, radix); np = add_chain(p, np, radix, nr, x, 1); }
    if (y > 1) { nr = global_radices(y, radix); np = add_chain(p, np, radix, nr, y, x); }
    if (z > 1) { nr = global_radices(z, radix); np = add_chain(p, np, radix, nr, z, x * y); }
    for (int i = 0; i < np; ++i) { R[i] = p[i].R; M[i] = p[i].M; S[i] = p[i].S; }
    return np;
}

#define DEFINE_PASS(T, NAME)                                                                              \
    static void NAME(const T* in, T* out, int64_t total, int R, int64_t M, int64_t S, int dir) {         \
        const int64_t outer = total / ((int64_t)R * M * S);                                               \
        T* wr = (T*)malloc(sizeof(T) * 2 * R);                                                            \
        for (int k = 0; k < R; ++k) {                                                                     \
            double a = dir * 2.0 * M_PI * (double)k / (double)R;                                          \
            wr[2 * k] = (T)cos(a);                                                                        \
            wr[2 * k + 1] = (T)sin(a);                                                                    \
        }                                                                                                 \
        const double tw_step = dir * 2.0 * M_PI / ((double)R * (double)M);                                \
        for (int64_t o = 0; o < outer; ++o)                                                               \
            for (int64_t l = 0; l < M; ++l)                                                               \
                for (int q = 0; q < R; ++q) {                                                             \
                    double ta = tw_step * (double)((l * (int64_t)q) % ((int64_t)R * M));                  \
                    const T tc = (T)cos(ta), ts = (T)sin(ta);                                             \
                    T* dst = out + 2 * (((o * M + l) * R + q) * S);                                       \
                    for (int64_t j = 0; j < S; ++j) { dst[2 * j] = 0; dst[2 * j + 1] = 0; }               \
                    for (int r = 0; r < R; ++r) {                                                         \
                        const int k = (int)(((int64_t)r * q) % R);                                        \
                        const T c = wr[2 * k], s = wr[2 * k + 1];                                         \
                        const T* src = in + 2 * (((o * R + r) * M + l) * S);                              \
                        for (int64_t j = 0; j < S; ++j) {                                                 \
                            const T xr = src[2 * j], xi = src[2 * j + 1];                                 \
                            dst[2 * j] += xr * c - xi * s;                                                \
                            dst[2 * j + 1] += xr * s + xi * c;                                            \
                        }                                                                                 \
                    }                                                                                     \
                    if (M > 1)                                                                            \
                        for (int64_t j = 0; j < S; ++j) {                                                 \
                            const T yr = dst[2 * j], yi = dst[2 * j + 1];                                 \
                            dst[2 * j] = yr * tc - yi * ts;                                               \
                            dst[2 * j + 1] = yr * ts + yi * tc;                                           \
                        }                                                                                 \
                }                                                                                         \
        free(wr);                                                                                         \
    }

DEFINE_PASS(float, pass_f32)
DEFINE_PASS(double, pass_f64)

/*
 * Execute the reference chain on `batch` transforms of shape (z, y, x), x contiguous, stored back to back.
 * inverse: 0 forward (exp(-...)), 1 inverse.  The last pass's outputs are divided by `divisor`
 * (kernel.py:23-37: forward 1/scale; inverse size*scale if normalize else scale).
 * Returns 0, or -1 on a bad argument.  `out` must not alias `in`.
 */
int oracle_execute(const void* in, void* out, int64_t x, int64_t y, int64_t z, int64_t batch, int prec, int inverse,
                   double divisor) {
    int R[MAX_PASSES];
    int64_t M[MAX_PASSES], S[MAX_PASSES];
    if (x < 1 || y < 1 || z < 1 || batch < 1) return -1;
    if ((x & (x - 1)) || (y & (y - 1)) || (z & (z - 1))) return -1;
    (void)ilog2;
    const int np = oracle_plan(x, y, z, prec, R, M, S);
    const int64_t total = x * y * z * batch;
    const size_t esz = prec ? 16 : 8;
    void* tmp = malloc(total * esz);
    if (!tmp) return -1;
    const void* cur = in;
    for (int i = 0; i < np; ++i) {
        /* ping-pong so that the last pass writes `out` */
        void* dst = ((np - 1 - i) % 2 == 0) ? out : tmp;
        if (prec) pass_f64((const double*)cur, (double*)dst, total, R[i], M[i], S[i], inverse ? 1 : -1);
        else pass_f32((const float*)cur, (float*)dst, total, R[i], M[i], S[i], inverse ? 1 : -1);
        cur = dst;
    }
    if (np == 0) memcpy(out, in, total * esz);
    if (divisor != 1.0) {
        if (prec) { double* d = (double*)out; for (int64_t i = 0; i < 2 * total; ++i) d[i] /= divisor; }
        else { float* d = (float*)out; const float dv = (float)divisor; for (int64_t i = 0; i < 2 * total; ++i) d[i] /= dv; }
    }
    free(tmp);
    return 0;
}
