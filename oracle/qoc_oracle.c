/*
 * qoc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY): plain-C restatement of the unitary-mode GRAPE iteration,
 * used (a) as a second, compiled checker of oracle/grape_oracle.py and (b) as the `cpu_baseline` leg of bench.py
 * ("port": our restatement of the reference's algorithm, not TensorFlow).  Never linked into libqoc_hip.so.
 *
 * Follows, per seed (reference file:line relative to /root/reference/quantum_optimal_control/):
 *   controls  u = maxA*sin(base)                               core/tensorflow_state.py:176-178
 *   K_t = (sum_{j<=T} A^j/j!)^(2^s), A = (H0' + sum u_k H_k')/2^s   core/tensorflow_state.py:25-46
 *   X_t = K_t X_{t-1},  Psi_t = K_t Psi_{t-1} (= X_t V)         core/tensorflow_state.py:204-242
 *   loss = 1 - |sum <w|psi>|^2/m^2, unitary_scale               core/tensorflow_state.py:225, 282-329
 *   dL/du_{k,t} = Re<Lambda_t, H_k' Psi_t>, Lambda_{t-1} = K_t^dagger Lambda_t   core/tensorflow_state.py:49-65
 *   grad = cos(base)*maxA*dL/du ; TF1 Adam                      core/tensorflow_state.py:342-356
 * Seeds are independent -> `#pragma omp parallel for` over seeds (the CPU analogue of the GPU's seed batch).
 * Regularisers are not restated here (bench workload C2 uses reg_coeffs = {}); grape_oracle.py covers them.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double _Complex cd;

static void mm(int M, int N, int K, const cd* A, const cd* B, cd* C) { /* C[MxN] = A[MxK] B[KxN] */
    for (int i = 0; i < M; ++i) {
        for (int j = 0; j < N; ++j) C[i * N + j] = 0;
        for (int k = 0; k < K; ++k) {
            const cd a = A[i * K + k];
            for (int j = 0; j < N; ++j) C[i * N + j] += a * B[k * N + j];
        }
    }
}
static void mm_dag(int n, int N, const cd* A, const cd* B, cd* C) { /* C[nxN] = A^dagger B, A n x n */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < N; ++j) C[i * N + j] = 0;
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n; ++i) {
            const cd a = conj(A[k * n + i]);
            for (int j = 0; j < N; ++j) C[i * N + j] += a * B[k * N + j];
        }
}

typedef struct {
    int n, k, steps, m, T, s;
} dims_t;

/* one evaluation of one seed; K, inter are scratch (steps*n*n, (steps+1)*n*m) */
static void eval_seed(const dims_t* d, const cd* Hs, const cd* U0, const cd* V, const cd* W, const double* maxA,
                      const double* base, double* loss, double* uscale, double* grad, cd* Ufinal, cd* K, cd* inter,
                      cd* tmp /* 4*n*n */) {
    const int n = d->n, k = d->k, steps = d->steps, m = d->m, nn = n * n, nm = n * m;
    cd *A = tmp, *Hn = tmp + nn, *Q = tmp + 2 * nn, *X = tmp + 3 * nn;
    const double inv = 1.0 / (double)(1 << d->s);
    memcpy(X, U0, sizeof(cd) * nn);
    memcpy(inter, V, sizeof(cd) * nm);
    cd* psi0 = (cd*)malloc(sizeof(cd) * nm);
    mm(n, m, n, U0, V, psi0);
    const cd* prev = psi0;
    for (int t = 0; t < steps; ++t) {
        cd* Kt = K + (size_t)t * nn;
        for (int o = 0; o < nn; ++o) {
            cd a = Hs[o] * inv;
            for (int kk = 0; kk < k; ++kk) a += (maxA[kk] * sin(base[kk * steps + t]) * inv) * Hs[(size_t)(kk + 1) * nn + o];
            A[o] = a; Hn[o] = a; Kt[o] = a;
        }
        for (int i = 0; i < n; ++i) Kt[i * n + i] += 1.0;
        double fact = 1.0;
        for (int ii = 2; ii <= d->T; ++ii) {
            mm(n, n, n, A, Hn, Q);
            fact *= ii;
            for (int o = 0; o < nn; ++o) { Hn[o] = Q[o]; Kt[o] += Q[o] / fact; }
        }
        for (int sq = 0; sq < d->s; ++sq) { mm(n, n, n, Kt, Kt, Q); memcpy(Kt, Q, sizeof(cd) * nn); }
        mm(n, n, n, Kt, X, Q); memcpy(X, Q, sizeof(cd) * nn);
        mm(n, m, n, Kt, prev, inter + (size_t)(t + 1) * nm);
        prev = inter + (size_t)(t + 1) * nm;
    }
    memcpy(Ufinal, X, sizeof(cd) * nn);
    double us = 0;
    for (int c = 0; c < n; ++c) { cd rs = 0; for (int a = 0; a < n; ++a) rs += X[c * n + a]; us += creal(rs * conj(rs)); }
    *uscale = us / n;
    cd z = 0;
    const cd* fin = inter + (size_t)steps * nm;
    for (int o = 0; o < nm; ++o) z += fin[o] * conj(W[o]);
    *loss = 1.0 - creal(z * conj(z)) / ((double)m * m);
    cd* lam = (cd*)malloc(sizeof(cd) * nm * 2);
    cd* lam2 = lam + nm;
    cd* Y = (cd*)malloc(sizeof(cd) * nm);
    for (int o = 0; o < nm; ++o) lam[o] = (-2.0 / ((double)m * m)) * z * W[o];
    for (int t = steps - 1; t >= 0; --t) {
        const cd* psi = inter + (size_t)(t + 1) * nm;
        for (int kk = 0; kk < k; ++kk) {
            mm(n, m, n, Hs + (size_t)(kk + 1) * nn, psi, Y);
            double g = 0;
            for (int o = 0; o < nm; ++o) g += creal(conj(lam[o]) * Y[o]);
            grad[kk * steps + t] = cos(base[kk * steps + t]) * maxA[kk] * g;
        }
        if (t == 0) break;
        mm_dag(n, m, K + (size_t)t * nn, lam, lam2);
        memcpy(lam, lam2, sizeof(cd) * nm);
    }
    free(lam); free(Y); free(psi0);
}

/* Evaluate n_seeds control sets; outputs per seed. */
int qoc_oracle_eval(int n, int k, int steps, int m, int T, int s, int n_seeds, const double* Hs, const double* U0,
                    const double* V, const double* W, const double* maxA, const double* base, double* loss,
                    double* uscale, double* grad, double* Ufinal, int nthreads) {
    dims_t d = {n, k, steps, m, T, s};
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, ks = (size_t)k * steps;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < n_seeds; ++b) {
        cd* K = (cd*)malloc(sizeof(cd) * nn * steps);
        cd* inter = (cd*)malloc(sizeof(cd) * nm * (steps + 1));
        cd* tmp = (cd*)malloc(sizeof(cd) * nn * 4);
        eval_seed(&d, (const cd*)Hs, (const cd*)U0, (const cd*)V, (const cd*)W, maxA, base + b * ks, loss + b,
                  uscale + b, grad + b * ks, (cd*)Ufinal + b * nn, K, inter, tmp);
        free(K); free(inter); free(tmp);
    }
    return 0;
}

/* `iters` full iterations (evaluate + TF1 Adam with lr = rate*exp(-it/decay)) of every seed; base updated in place. */
int qoc_oracle_iterate(int n, int k, int steps, int m, int T, int s, int n_seeds, const double* Hs, const double* U0,
                       const double* V, const double* W, const double* maxA, double* base, int iters, double rate,
                       double decay, double* loss_out, int nthreads) {
    dims_t d = {n, k, steps, m, T, s};
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, ks = (size_t)k * steps;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < n_seeds; ++b) {
        cd* K = (cd*)malloc(sizeof(cd) * nn * steps);
        cd* inter = (cd*)malloc(sizeof(cd) * nm * (steps + 1));
        cd* tmp = (cd*)malloc(sizeof(cd) * nn * 4);
        cd* Uf = (cd*)malloc(sizeof(cd) * nn);
        double* g = (double*)malloc(sizeof(double) * ks);
        double* am = (double*)calloc(ks, sizeof(double));
        double* av = (double*)calloc(ks, sizeof(double));
        double* x = base + b * ks;
        double us;
        for (int it = 1; it <= iters; ++it) {
            eval_seed(&d, (const cd*)Hs, (const cd*)U0, (const cd*)V, (const cd*)W, maxA, x, loss_out + b, &us, g, Uf, K,
                      inter, tmp);
            const double lr = rate * exp(-(double)it / decay);
            const double lr_t = lr * sqrt(1.0 - pow(0.999, it)) / (1.0 - pow(0.9, it));
            for (size_t o = 0; o < ks; ++o) {
                am[o] = 0.9 * am[o] + 0.1 * g[o];
                av[o] = 0.999 * av[o] + 0.001 * g[o] * g[o];
                x[o] -= lr_t * am[o] / (sqrt(av[o]) + 1e-8);
            }
        }
        free(K); free(inter); free(tmp); free(Uf); free(g); free(am); free(av);
    }
    return 0;
}

int qoc_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
