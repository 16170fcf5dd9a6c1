// EXPERIMENT (round 4; CPU only): can the Filter scan of jit_prelude.hpp JitFilterScan be made BIT-EQUAL by "verify and repair"?
//
// The proposal (DESIGN.md §9 item 1 (iii) of round 3, the round-3 review's item 1a): after the linear scan every lane re-runs its samples
// with the reference's roundings from the pair the scan handed it; then check per lane that the pair it entered with equals the pair the lane
// below left with AFTER its f32 roundings, and re-run the lanes that fail — exact by induction from lane 0.  Each repair round moves the
// exact frontier by one lane's samples (L) unless the trajectories have MERGED before: a recurrence started from a pair that is a few ulps
// off (the scan's pairs lack the reference's rounding errors) follows the reference's trajectory at a distance, and is bit-equal from the
// first sample on at which two consecutive outputs coincide.  How soon that happens decides the cost.  This program measures it:
// Filter.js:40-46 as written (f64, f32 store per step) over ten seconds of a configs[3]-like input; per 256-sample chunk the unrounded
// recurrence from the exact pair at the chunk's start gives the scan's pair at every slice start; from each the exact recurrence is run
// until it merges.  Printed: the distribution of the merge time in samples, and the repair rounds a chunk needs (the slowest of its slices
// decides: the wave loops until every lane verifies).
//
// Result (profiles/r04_filter_merge.txt): at configs[3]'s 2 kHz the quantised recurrence has a dead band — its all-pole part has a DC gain of
// 18.5, so two trajectories one ulp apart stay one ulp apart with probability ~0.95 per step — the mean merge time is 29 samples, 14 % of
// the slices need more than 64 and 2 % more than 128; a chunk needs 30 repair rounds on average at L = 4 (76 at worst), i.e. ~122 exact
// steps per lane where the serial recurrence needs 256 on ONE lane per instance: ~1000 instructions per chunk and wavefront against the
// scan's 115 and the stage's 64.  At 4 kHz 9 rounds, at 8 kHz 3.  Verify-and-repair is therefore NOT built: below ~6 kHz it costs more than
// the Filter stage it would replace.
//   gcc -O2 -ffp-contract=off -o /tmp/filter_merge tools/filter_merge_experiment.c -lm && /tmp/filter_merge 2000 4
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static void coef(double f, double sr, double *k) {
    double lam = 1 / tan(M_PI * f / sr);
    double a0 = 1 / (1 + 2 * lam + lam * lam);
    k[0] = a0; k[1] = 2 * a0; k[2] = a0; k[3] = 2 * a0 * (1 - lam * lam); k[4] = a0 * (1 - 2 * lam + lam * lam);
}
// --from-rest <cutoff> [saw]: the other question — a recurrence started from REST (zero pair, zero input history) somewhere in a long signal:
// after how many samples is it the sequential run's, bit for bit?  What a time segment of ONE filtered circuit needs as warm-up
// (jit_codegen.hpp jit_warm_chunks: 32 sum|h| samples are given; profiles/r04_filter_merge.txt: every trial merged within 20 sum|h| down to 400 Hz,
// and at 200 Hz — sum|h| 1500 — effectively never: the quantised recurrence has limit cycles there, so such cutoffs are not cut at all).
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
static int from_rest(double fc, int saw) {
    const double sr = 48000;
    double k[5];
    coef(fc, sr, k);
    const int N = 48000 * 40;
    float *x = malloc(N * 4), *y = malloc(N * 4);
    for (int t = 0; t < N; t++) x[t] = saw ? (float)(2.0 * fmod(110.0 * t / sr, 1.0) - 1.0) : (float)sin(2 * M_PI * 110.0 * t / sr);
    double y1 = 0, y2 = 0, x1 = 0, x2 = 0;
    for (int t = 0; t < N; t++) {
        double P = (k[0] * x[t] + k[1] * x1) + k[2] * x2;
        float yy = (float)((P - k[3] * y1) - k[4] * y2);
        y[t] = yy; y2 = y1; y1 = yy; x2 = x1; x1 = x[t];
    }
    int trials = 4000, *m = malloc(trials * sizeof(int));
    for (int i = 0; i < trials; i++) {
        int t0 = 48000 + (int)((double)rand() / RAND_MAX * (N - 48000 * 12));
        double u1 = 0, u2 = 0, a1 = 0, a2 = 0;
        int same = 0, steps = 0;
        for (int q = t0; q < N; q++) {
            double P = (k[0] * x[q] + k[1] * a1) + k[2] * a2;
            float yy = (float)((P - k[3] * u1) - k[4] * u2);
            u2 = u1; u1 = yy; a2 = a1; a1 = x[q]; steps++;
            if (yy == y[q]) { if (++same == 2) break; } else same = 0;
        }
        m[i] = steps;
    }
    qsort(m, trials, sizeof(int), cmp_int);
    double lam = 1 / tan(M_PI * fc / sr), sum = (1 + lam) * (1 + lam) / 4;
    printf("from rest, fc %6.0f %s sum|h| %8.1f: merged after  median %7d  90%% %7d  99%% %7d  99.9%% %7d  max %7d samples  (max / sum|h| = %.1f)\n", fc, saw ? "saw " : "sine", sum,
           m[trials / 2], m[trials * 9 / 10], m[trials * 99 / 100], m[trials * 999 / 1000], m[trials - 1], m[trials - 1] / sum);
    return 0;
}
int main(int argc, char **argv) {
    if (argc > 2 && !strcmp(argv[1], "--from-rest")) return from_rest(atof(argv[2]), argc > 3);
    double fc = argc > 1 ? atof(argv[1]) : 2000, sr = 48000;
    int L = argc > 2 ? atoi(argv[2]) : 4;
    double k[5]; coef(fc, sr, k);
    const int N = 480000;
    float *x = malloc(N * 4), *y = malloc(N * 4);
    double *P = malloc(N * 8);
    // input: a sine at 130.37 Hz plus its half-gain echo (roughly configs[3]'s signal scale)
    for (int t = 0; t < N; t++) x[t] = (float)(sin(2 * M_PI * 130.37 * t / sr) + 0.5 * sin(2 * M_PI * 130.37 * (t - 480) / sr + 0.3));
    double y1 = 0, y2 = 0; float x1 = 0, x2 = 0;
    for (int t = 0; t < N; t++) {
        P[t] = (k[0] * x[t] + k[1] * x1) + k[2] * x2;
        float yy = (float)((P[t] - k[3] * y1) - k[4] * y2);
        y[t] = yy; y2 = y1; y1 = yy; x2 = x1; x1 = x[t];
    }
    // per chunk of 256: linear (unrounded, fma) run from the exact pair at chunk start; at every slice start s*L take the linear pair as the guess,
    // run exactly from there, count steps until two consecutive outputs equal the reference's
    long hist[4096] = {0}; long n = 0; double sum = 0; long over[6] = {0}; int lim[6] = {8, 16, 32, 64, 128, 256};
    long wave_rounds_hist[300] = {0}; long waves = 0;
    for (int c0 = 256; c0 + 256 + 2048 < N; c0 += 256) {
        double c1 = y[c0 - 1], c2 = y[c0 - 2];
        int worst = 0;
        for (int t = c0; t < c0 + 256; t++) {
            if ((t - c0) % L == 0 && t > c0) {
                double u1 = c1, u2 = c2; int m = 0, same = 0;
                for (int q = t; q < t + 2000; q++) {
                    float yy = (float)((P[q] - k[3] * u1) - k[4] * u2);
                    u2 = u1; u1 = yy; m++;
                    if (yy == y[q]) { if (++same == 2) break; } else same = 0;
                }
                m -= 2;  // steps before the first of the two equal outputs
                hist[m < 4095 ? m : 4095]++; n++; sum += m;
                for (int i = 0; i < 6; i++) if (m > lim[i]) over[i]++;
                if (m > worst) worst = m;
            }
            double z = fma(-k[4], c2, fma(-k[3], c1, P[t]));
            c2 = c1; c1 = z;
        }
        int rounds = 1 + (worst + L - 1) / L;  // Jacobi rounds of L samples until every slice of the chunk entered exactly (+1 to verify)
        wave_rounds_hist[rounds < 299 ? rounds : 299]++; waves++;
    }
    printf("fc %.0f L %d: slices %ld mean merge %.1f samples;", fc, L, n, sum / n);
    for (int i = 0; i < 6; i++) printf(" >%d: %.3f%%", lim[i], 100.0 * over[i] / n);
    double mr = 0; int maxr = 0;
    for (int r = 0; r < 300; r++) { mr += (double)r * wave_rounds_hist[r]; if (wave_rounds_hist[r]) maxr = r; }
    printf("\n  rounds per chunk (max over its slices): mean %.1f max %d -> exact steps per lane %.0f (serial: 256)\n", mr / waves, maxr, mr / waves * L);
    return 0;
}
