/*
 * CPU ORACLE (test infrastructure - NOT product code): plain-C restatement of the ProbSparse window
 * attention core, forward and backward.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product never does.
 *
 * Follows the reference's ProbAttention (Uformer_ProbSparse/ProbSparse/attn.py):
 *   forward  ATT:287-342  = _prob_QK ATT:71-152 (sampled scores ATT:104-110, M = max - sum/L_K ATT:117,
 *            top-u ATT:122, Q_reduce K^T ATT:150), scale ATT:327-329, mean(V) context ATT:168-172,
 *            softmax ATT:195, + bias[h, top] ATT:229, + mask[b % nW, top] ATT:251-258, softmax ATT:262,
 *            context[top] = A V ATT:271-272.
 *   backward = the autograd of exactly those steps (no gradient through sampling / top-u selection).
 * Arithmetic is carried in double so that this oracle is an independent, higher-precision check of
 * both the torch restatement (oracle/uformer_oracle.py) and the HIP kernel.
 * Parity status: PINNED - tests/test_oracle_golden.py checks it against the golden vectors captured
 * from the reference (tests/golden/probattn_*.npz).
 *
 * Layout: q,k,v,ctx and their gradients are [B_, H, N, d] contiguous fp32.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void softmax_row(const double* x, double* p, int n) {
    double mx = x[0], s = 0.0;
    for (int j = 1; j < n; ++j) if (x[j] > mx) mx = x[j];
    for (int j = 0; j < n; ++j) { p[j] = exp(x[j] - mx); s += p[j]; }
    for (int j = 0; j < n; ++j) p[j] /= s;
}

/* top[b,h,0..u-1]: selected queries ordered by descending M (ties: lower index first). */
int ps_attn_oracle_fwd(const float* q, const float* k, const float* v, const int* idx, const float* bias,
                       const float* mask, int B_, int H, int nW, int N, int d, int u, float* ctx, int* top,
                       float* Mout) {
    double* S = (double*)malloc(sizeof(double) * N * N);
    double* M = (double*)malloc(sizeof(double) * N);
    double* a = (double*)malloc(sizeof(double) * N);
    double* p = (double*)malloc(sizeof(double) * N);
    char* used = (char*)malloc(N);
    const double scale = 1.0 / sqrt((double)d);
    for (int b = 0; b < B_; ++b)
        for (int h = 0; h < H; ++h) {
            const float* Q = q + ((size_t)(b * H + h)) * N * d;
            const float* K = k + ((size_t)(b * H + h)) * N * d;
            const float* V = v + ((size_t)(b * H + h)) * N * d;
            float* C = ctx + ((size_t)(b * H + h)) * N * d;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    double s = 0.0;
                    for (int e = 0; e < d; ++e) s += (double)Q[i * d + e] * (double)K[j * d + e];
                    S[i * N + j] = s;
                }
            for (int i = 0; i < N; ++i) {                       /* sparsity measure over the SAMPLED keys */
                double mx = -INFINITY, sum = 0.0;
                for (int s = 0; s < u; ++s) {
                    const double val = S[i * N + idx[i * u + s]];
                    if (val > mx) mx = val;
                    sum += val;
                }
                M[i] = mx - sum / (double)N;
                if (Mout) Mout[(size_t)(b * H + h) * N + i] = (float)M[i];
            }
            memset(used, 0, N);
            int* T = top + ((size_t)(b * H + h)) * u;
            for (int r = 0; r < u; ++r) {                       /* selection by repeated arg-max */
                int best = -1;
                for (int i = 0; i < N; ++i)
                    if (!used[i] && (best < 0 || M[i] > M[best])) best = i;
                used[best] = 1;
                T[r] = best;
            }
            for (int e = 0; e < d; ++e) {                       /* context initialised to mean(V) */
                double s = 0.0;
                for (int j = 0; j < N; ++j) s += (double)V[j * d + e];
                for (int i = 0; i < N; ++i) C[i * d + e] = (float)(s / (double)N);
            }
            for (int r = 0; r < u; ++r) {
                const int i = T[r];
                for (int j = 0; j < N; ++j) a[j] = S[i * N + j] * scale;
                softmax_row(a, p, N);
                for (int j = 0; j < N; ++j) {
                    a[j] = p[j];
                    if (bias) a[j] += (double)bias[((size_t)h * N + i) * N + j];
                    if (mask) a[j] += (double)mask[((size_t)(b % nW) * N + i) * N + j];
                }
                softmax_row(a, p, N);
                for (int e = 0; e < d; ++e) {
                    double s = 0.0;
                    for (int j = 0; j < N; ++j) s += p[j] * (double)V[j * d + e];
                    C[i * d + e] = (float)s;
                }
            }
        }
    free(S); free(M); free(a); free(p); free(used);
    return 0;
}

/* dbias: [H,N,N], ACCUMULATED over windows (caller zeroes), may be NULL. */
int ps_attn_oracle_bwd(const float* q, const float* k, const float* v, const float* bias, const float* mask,
                       const int* top, const float* dctx, int B_, int H, int nW, int N, int d, int u, float* dq,
                       float* dk, float* dv, float* dbias) {
    double* x = (double*)malloc(sizeof(double) * N);
    double* p1 = (double*)malloc(sizeof(double) * N);
    double* p2 = (double*)malloc(sizeof(double) * N);
    double* dp = (double*)malloc(sizeof(double) * N);
    double* dKa = (double*)malloc(sizeof(double) * N * d);
    double* dVa = (double*)malloc(sizeof(double) * N * d);
    char* sel = (char*)malloc(N);
    const double scale = 1.0 / sqrt((double)d);
    for (int b = 0; b < B_; ++b)
        for (int h = 0; h < H; ++h) {
            const size_t o = ((size_t)(b * H + h)) * N * d;
            const float *Q = q + o, *K = k + o, *V = v + o, *dC = dctx + o;
            float *dQ = dq + o, *dK = dk + o, *dV = dv + o;
            const int* T = top + ((size_t)(b * H + h)) * u;
            memset(sel, 0, N);
            for (int r = 0; r < u; ++r) sel[T[r]] = 1;
            for (int i = 0; i < N * d; ++i) { dKa[i] = 0.0; dVa[i] = 0.0; dQ[i] = 0.f; }
            for (int e = 0; e < d; ++e) {                       /* mean(V) path: unselected rows */
                double s = 0.0;
                for (int i = 0; i < N; ++i) if (!sel[i]) s += (double)dC[i * d + e];
                for (int j = 0; j < N; ++j) dVa[j * d + e] += s / (double)N;
            }
            for (int r = 0; r < u; ++r) {
                const int i = T[r];
                for (int j = 0; j < N; ++j) {
                    double s = 0.0;
                    for (int e = 0; e < d; ++e) s += (double)Q[i * d + e] * (double)K[j * d + e];
                    x[j] = s * scale;
                }
                softmax_row(x, p1, N);
                for (int j = 0; j < N; ++j) {
                    x[j] = p1[j];
                    if (bias) x[j] += (double)bias[((size_t)h * N + i) * N + j];
                    if (mask) x[j] += (double)mask[((size_t)(b % nW) * N + i) * N + j];
                }
                softmax_row(x, p2, N);
                double dot = 0.0;
                for (int j = 0; j < N; ++j) {                   /* dP2 = dC_i . V_j ; dV += P2^T dC */
                    double s = 0.0;
                    for (int e = 0; e < d; ++e) {
                        s += (double)dC[i * d + e] * (double)V[j * d + e];
                        dVa[j * d + e] += p2[j] * (double)dC[i * d + e];
                    }
                    dp[j] = s;
                    dot += s * p2[j];
                }
                double dot1 = 0.0;
                for (int j = 0; j < N; ++j) {                   /* softmax-2 backward -> dA (= dbias row = dP1) */
                    dp[j] = p2[j] * (dp[j] - dot);
                    if (dbias) dbias[((size_t)h * N + i) * N + j] += (float)dp[j];
                    dot1 += dp[j] * p1[j];
                }
                for (int j = 0; j < N; ++j) {                   /* softmax-1 backward, scale */
                    const double ds = p1[j] * (dp[j] - dot1) * scale;
                    for (int e = 0; e < d; ++e) {
                        dQ[i * d + e] += (float)0.0;            /* accumulated in double below */
                        dKa[j * d + e] += ds * (double)Q[i * d + e];
                    }
                    dp[j] = ds;
                }
                for (int e = 0; e < d; ++e) {
                    double s = 0.0;
                    for (int j = 0; j < N; ++j) s += dp[j] * (double)K[j * d + e];
                    dQ[i * d + e] = (float)s;
                }
            }
            for (int i = 0; i < N * d; ++i) { dK[i] = (float)dKa[i]; dV[i] = (float)dVa[i]; }
        }
    free(x); free(p1); free(p2); free(dp); free(dKa); free(dVa); free(sel);
    return 0;
}
