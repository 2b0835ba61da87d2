/*
 * rgb2spec_opt.c -- generator for data/srgb.coeff, the sigmoid-polynomial spectral-upsampling table the
 * reference loads in UpsampleTable::load (src/spectrum/upsample.rs:108-144) and which is missing from the
 * reference tree (.MISSING_LARGE_BLOBS).
 *
 * It restates the published optimiser of Jakob & Hanika, "A Low-Dimensional Function Space for Efficient
 * Spectral Upsampling" (EG 2019; the reference cites it at upsample.rs:1): for every cell of a
 * 3 x res^3 grid over (max channel, other two channels scaled by it, brightness) find by Gauss-Newton the
 * three coefficients whose sigmoid(polynomial) reflectance spectrum, lit by D65 and seen through the reference's
 * own observer, reproduces the cell's sRGB colour in CIE Lab.  Data: every fifth entry of the reference's 1 nm
 * colour-matching tables (color.rs:145-1565; by their values the CIE 1964 10-degree observer -- NOT the CIE 1931
 * 2-degree observer that Jakob & Hanika's published tool, and hence the reference's missing data/srgb.coeff, used,
 * which is why the golden vector of upsample.rs:175-185 is not reproduced) and the CIE D65 illuminant
 * (published 10 nm values, midpoints in between, as the 5 nm CIE table has them).
 *
 *   rgb2spec_opt <resolution> <out-file>                  full table (resolution 64 = the reference's file)
 *   rgb2spec_opt --cell <resolution> <l> <j> <i> <k>      print the coefficients of one cell
 *
 * File format (upsample.rs:108-144): "SPEC", u32 LE resolution, resolution f32 LE scale values, then
 * 3 * resolution^3 * 3 f32 LE coefficients indexed (((l*res + k)*res + j)*res + i)*3 (upsample.rs:76).
 *
 * Build: gcc -O2 -fopenmp -o rgb2spec_opt tools/rgb2spec_opt.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define IRIS_TABLE_QUAL const
#include "../oracle/tables/cie_cmf.inc"

#define LAMBDA_MIN 360.0
#define LAMBDA_MAX 830.0
#define CIE_SAMPLES 95
#define FINE_SAMPLES ((CIE_SAMPLES - 1) * 3 + 1)
#define EPSILON 1e-4

/* CIE standard illuminant D65, 360..830 nm in 10 nm steps */
static const double D65_10NM[48] = {
    46.6383, 52.0891, 49.9755, 54.6482, 82.7549, 91.486,  93.4318, 86.6823, 104.865, 117.008, 117.812, 114.861,
    115.923, 108.811, 109.354, 107.802, 104.79,  107.689, 104.405, 104.046, 100.0,   96.3342, 95.788,  88.6856,
    90.0062, 89.5991, 87.6987, 83.2886, 83.6992, 80.0268, 80.2146, 82.2778, 78.2842, 69.7213, 71.6091, 74.349,
    61.604,  69.8856, 75.087,  63.5927, 46.4182, 66.8054, 63.3828, 64.304,  59.4519, 51.959,  57.4406, 60.3125};

static const double XYZ_TO_SRGB[3][3] = {{3.240479, -1.537150, -0.498535}, {-0.969256, 1.875991, 0.041556}, {0.055648, -0.204043, 1.057311}};
static const double SRGB_TO_XYZ[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};

static double cie_x[CIE_SAMPLES], cie_y[CIE_SAMPLES], cie_z[CIE_SAMPLES], d65[CIE_SAMPLES];
static double lambda_tbl[FINE_SAMPLES], rgb_tbl[3][FINE_SAMPLES], xyz_whitepoint[3];

static double sqr(double x) { return x * x; }
static double sigmoid(double x) { return 0.5 * x / sqrt(1.0 + x * x) + 0.5; }
static double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }

static double cie_interp(const double *data, double x) {
    x -= LAMBDA_MIN;
    x *= (CIE_SAMPLES - 1) / (LAMBDA_MAX - LAMBDA_MIN);
    int offset = (int)x;
    if (offset < 0) offset = 0;
    if (offset > CIE_SAMPLES - 2) offset = CIE_SAMPLES - 2;
    double weight = x - offset;
    return (1.0 - weight) * data[offset] + weight * data[offset + 1];
}

static void init_tables(void) {
    for (int i = 0; i < CIE_SAMPLES; i++) {
        cie_x[i] = IRIS_CIE_X[5 * i];
        cie_y[i] = IRIS_CIE_Y[5 * i];
        cie_z[i] = IRIS_CIE_Z[5 * i];
        d65[i] = (i % 2 == 0) ? D65_10NM[i / 2] : 0.5 * (D65_10NM[i / 2] + D65_10NM[i / 2 + 1]);
    }
    /* normalise the illuminant so that the white point has Y = 1 under the quadrature used below */
    const double h = (LAMBDA_MAX - LAMBDA_MIN) / (FINE_SAMPLES - 1);
    double weights[FINE_SAMPLES], y_white = 0.0;
    for (int i = 0; i < FINE_SAMPLES; i++) {
        lambda_tbl[i] = LAMBDA_MIN + i * h;
        double w = 3.0 / 8.0 * h; /* Simpson 3/8 */
        if (i == 0 || i == FINE_SAMPLES - 1) ;
        else if ((i - 1) % 3 == 2) w *= 2.0;
        else w *= 3.0;
        weights[i] = w;
        y_white += cie_interp(cie_y, lambda_tbl[i]) * cie_interp(d65, lambda_tbl[i]) * w;
    }
    memset(rgb_tbl, 0, sizeof rgb_tbl);
    memset(xyz_whitepoint, 0, sizeof xyz_whitepoint);
    for (int i = 0; i < FINE_SAMPLES; i++) {
        const double lambda = lambda_tbl[i];
        const double xyz[3] = {cie_interp(cie_x, lambda), cie_interp(cie_y, lambda), cie_interp(cie_z, lambda)};
        const double I = cie_interp(d65, lambda) / y_white;
        for (int k = 0; k < 3; k++)
            for (int j = 0; j < 3; j++) rgb_tbl[k][i] += XYZ_TO_SRGB[k][j] * xyz[j] * I * weights[i];
        for (int k = 0; k < 3; k++) xyz_whitepoint[k] += xyz[k] * I * weights[i];
    }
}

static void cie_lab(double *p) {
    double X = 0.0, Y = 0.0, Z = 0.0;
    const double Xw = xyz_whitepoint[0], Yw = xyz_whitepoint[1], Zw = xyz_whitepoint[2];
    for (int j = 0; j < 3; j++) {
        X += p[j] * SRGB_TO_XYZ[0][j];
        Y += p[j] * SRGB_TO_XYZ[1][j];
        Z += p[j] * SRGB_TO_XYZ[2][j];
    }
#define LAB_F(t) (((t) > (6.0 / 29.0) * (6.0 / 29.0) * (6.0 / 29.0)) ? cbrt(t) : (t) / (3.0 * (6.0 / 29.0) * (6.0 / 29.0)) + 4.0 / 29.0)
    const double fx = LAB_F(X / Xw), fy = LAB_F(Y / Yw), fz = LAB_F(Z / Zw);
    p[0] = 116.0 * fy - 16.0;
    p[1] = 500.0 * (fx - fy);
    p[2] = 200.0 * (fy - fz);
}

static void eval_residual(const double *coeffs, const double *rgb, double *residual) {
    double out[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < FINE_SAMPLES; i++) {
        const double lambda = (lambda_tbl[i] - LAMBDA_MIN) / (LAMBDA_MAX - LAMBDA_MIN);
        double x = 0.0;
        for (int k = 0; k < 3; k++) x = x * lambda + coeffs[k];
        const double s = sigmoid(x);
        for (int j = 0; j < 3; j++) out[j] += rgb_tbl[j][i] * s;
    }
    cie_lab(out);
    memcpy(residual, rgb, 3 * sizeof(double));
    cie_lab(residual);
    for (int j = 0; j < 3; j++) residual[j] -= out[j];
}

static void eval_jacobian(const double *coeffs, const double *rgb, double jac[3][3]) {
    double r0[3], r1[3], tmp[3];
    for (int i = 0; i < 3; i++) {
        memcpy(tmp, coeffs, sizeof tmp);
        tmp[i] -= EPSILON;
        eval_residual(tmp, rgb, r0);
        memcpy(tmp, coeffs, sizeof tmp);
        tmp[i] += EPSILON;
        eval_residual(tmp, rgb, r1);
        for (int j = 0; j < 3; j++) jac[j][i] = (r1[j] - r0[j]) * 1.0 / (2.0 * EPSILON);
    }
}

/* 3x3 LU decomposition with partial pivoting, then solve */
static int lu_solve(double A[3][3], const double *b, double *x) {
    int P[3] = {0, 1, 2};
    for (int i = 0; i < 3; i++) {
        double max_a = 0.0;
        int imax = i;
        for (int k = i; k < 3; k++)
            if (fabs(A[k][i]) > max_a) { max_a = fabs(A[k][i]); imax = k; }
        if (max_a < 1e-15) return 0;
        if (imax != i) {
            int t = P[i]; P[i] = P[imax]; P[imax] = t;
            for (int c = 0; c < 3; c++) { double tt = A[i][c]; A[i][c] = A[imax][c]; A[imax][c] = tt; }
        }
        for (int j = i + 1; j < 3; j++) {
            A[j][i] /= A[i][i];
            for (int k = i + 1; k < 3; k++) A[j][k] -= A[j][i] * A[i][k];
        }
    }
    for (int i = 0; i < 3; i++) {
        x[i] = b[P[i]];
        for (int k = 0; k < i; k++) x[i] -= A[i][k] * x[k];
    }
    for (int i = 2; i >= 0; i--) {
        for (int k = i + 1; k < 3; k++) x[i] -= A[i][k] * x[k];
        x[i] /= A[i][i];
    }
    return 1;
}

static int gauss_newton(const double rgb[3], double coeffs[3]) {
    for (int it = 0; it < 15; it++) {
        double J[3][3], residual[3], x[3];
        eval_residual(coeffs, rgb, residual);
        eval_jacobian(coeffs, rgb, J);
        if (!lu_solve(J, residual, x)) return 0;
        double r = 0.0;
        for (int j = 0; j < 3; j++) {
            coeffs[j] -= x[j];
            r += sqr(residual[j]);
        }
        const double max = fmax(fmax(coeffs[0], coeffs[1]), coeffs[2]);
        if (max > 200.0)
            for (int j = 0; j < 3; j++) coeffs[j] *= 200.0 / max;
        if (r < 1e-6) break;
    }
    return 1;
}

static void store(float *out, int res, int l, int k, int j, int i, const double coeffs[3]) {
    const double c0 = LAMBDA_MIN, c1 = 1.0 / (LAMBDA_MAX - LAMBDA_MIN);
    const double A = coeffs[0], B = coeffs[1], C = coeffs[2];
    const size_t idx = (((size_t)l * res + k) * res + j) * res + i;
    out[3 * idx + 0] = (float)(A * sqr(c1));
    out[3 * idx + 1] = (float)(B * c1 - 2.0 * A * c0 * sqr(c1));
    out[3 * idx + 2] = (float)(C - B * c0 * c1 + A * sqr(c0 * c1));
}

/* one (l, j, i) column of the table: brightness index k sweeps up and down from res/5 with warm starts */
static int solve_column(float *out, const float *scale, int res, int l, int j, int i) {
    const double y = (double)j / (res - 1), x = (double)i / (res - 1);
    const int start = res / 5;
    double coeffs[3] = {0.0, 0.0, 0.0}, rgb[3];
    for (int k = start; k < res; k++) {
        const double b = (double)scale[k];
        rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
        if (!gauss_newton(rgb, coeffs)) return 0;
        store(out, res, l, k, j, i, coeffs);
    }
    memset(coeffs, 0, sizeof coeffs);
    for (int k = start; k >= 0; k--) {
        const double b = (double)scale[k];
        rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
        if (!gauss_newton(rgb, coeffs)) return 0;
        store(out, res, l, k, j, i, coeffs);
    }
    return 1;
}

int main(int argc, char **argv) {
    init_tables();
    if (argc == 7 && strcmp(argv[1], "--cell") == 0) {
        const int res = atoi(argv[2]), l = atoi(argv[3]), j = atoi(argv[4]), i = atoi(argv[5]), k = atoi(argv[6]);
        if (res < 5 || l < 0 || l > 2 || j < 0 || j >= res || i < 0 || i >= res || k < 0 || k >= res) return 2;
        float *scale = malloc(sizeof(float) * res), *out = calloc((size_t)9 * res * res * res, sizeof(float));
        for (int t = 0; t < res; t++) scale[t] = (float)smoothstep(smoothstep(t / (double)(res - 1)));
        if (!solve_column(out, scale, res, l, j, i)) return 1;
        const size_t idx = (((size_t)l * res + k) * res + j) * res + i;
        printf("%.9g %.9g %.9g\n", out[3 * idx], out[3 * idx + 1], out[3 * idx + 2]);
        return 0;
    }
    if (argc != 3) {
        fprintf(stderr, "usage: %s <resolution> <out-file> | --cell <resolution> <l> <j> <i> <k>\n", argv[0]);
        return 2;
    }
    const int res = atoi(argv[1]);
    if (res < 5 || res > 256) return 2;
    float *scale = malloc(sizeof(float) * res);
    for (int k = 0; k < res; k++) scale[k] = (float)smoothstep(smoothstep(k / (double)(res - 1)));
    const size_t n = (size_t)9 * res * res * res;
    float *out = calloc(n, sizeof(float));
    int failed = 0;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int l = 0; l < 3; l++)
        for (int j = 0; j < res; j++)
            for (int i = 0; i < res; i++)
                if (!solve_column(out, scale, res, l, j, i)) failed = 1;
    if (failed) { fprintf(stderr, "Gauss-Newton: singular Jacobian\n"); return 1; }
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 1;
    const uint32_t r32 = (uint32_t)res;
    fwrite("SPEC", 4, 1, f);
    fwrite(&r32, 4, 1, f);
    fwrite(scale, sizeof(float), res, f);
    fwrite(out, sizeof(float), n, f);
    return fclose(f) == 0 ? 0 : 1;
}
