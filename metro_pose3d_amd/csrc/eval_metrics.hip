// Evaluation metrics on the device -- the step AFTER the path.
// Restates reference src/main.py:339-359 (build_eval_metrics):
//   dist[n,j]    = || root_relative(pred - true) ||          (root = LAST joint, tfu3d.py:23-25)
//   dist_pa[n,j] = the same after rigid alignment with scale, no reflection
//                  (tfu3d.rigid_align -> util3d.py:139-159 -> eval/procrustes.py:6-107)
//   auc score    = max(0, 1 - dist/150), pck = dist <= 150, masked means per joint and overall
//                  (tfu.reduce_mean_masked, tfu.py:44-66)
// One thread per pose does the Procrustes fit in fp64: centre and scale both point sets over the
// valid joints, A = X0^T Y0, SVD by cyclic Jacobi on A^T A, T = V U^T with the last singular
// direction flipped when det(T) < 0.  A second kernel reduces over poses per joint (fp64 sums).
#include "metro_common.h"

namespace metro {

namespace {

__device__ void jacobi_eig3(double b[3][3], double v[3][3], double lam[3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(b[0][1]) + fabs(b[0][2]) + fabs(b[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(b[p][q]) < 1e-300) continue;
                const double theta = (b[q][q] - b[p][p]) / (2.0 * b[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {          // B <- J^T B J
                    const double bkp = b[k][p], bkq = b[k][q];
                    b[k][p] = c * bkp - s * bkq;
                    b[k][q] = s * bkp + c * bkq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double bpk = b[p][k], bqk = b[q][k];
                    b[p][k] = c * bpk - s * bqk;
                    b[q][k] = s * bpk + c * bqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) lam[i] = b[i][i];
}

__device__ double det3(const double m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

}  // namespace

__global__ __launch_bounds__(64) void eval_pose_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                       const unsigned char* __restrict__ valid, int n, int nj,
                                                       float* __restrict__ dist, float* __restrict__ dist_pa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* P = pred + (size_t)i * nj * 3;
    const float* G = truth + (size_t)i * nj * 3;
    const unsigned char* M = valid + (size_t)i * nj;
    // --- plain root-relative distance
    double rd[3];
    for (int c = 0; c < 3; ++c) rd[c] = (double)P[(nj - 1) * 3 + c] - (double)G[(nj - 1) * 3 + c];
    for (int j = 0; j < nj; ++j) {
        double s = 0;
        for (int c = 0; c < 3; ++c) {
            const double d = ((double)P[j * 3 + c] - (double)G[j * 3 + c]) - rd[c];
            s += d * d;
        }
        dist[(size_t)i * nj + j] = (float)sqrt(s);
    }
    // --- Procrustes over the valid joints: X = truth, Y = pred (procrustes.py:40-58)
    double mux[3] = {0, 0, 0}, muy[3] = {0, 0, 0};
    int cnt = 0;
    for (int j = 0; j < nj; ++j)
        if (M[j]) {
            ++cnt;
            for (int c = 0; c < 3; ++c) { mux[c] += G[j * 3 + c]; muy[c] += P[j * 3 + c]; }
        }
    for (int c = 0; c < 3; ++c) { mux[c] /= cnt; muy[c] /= cnt; }
    double ssx = 0, ssy = 0, A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < nj; ++j)
        if (M[j]) {
            double x0[3], y0[3];
            for (int c = 0; c < 3; ++c) { x0[c] = G[j * 3 + c] - mux[c]; y0[c] = P[j * 3 + c] - muy[c]; ssx += x0[c] * x0[c]; ssy += y0[c] * y0[c]; }
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) A[r][c] += x0[r] * y0[c];
        }
    const double normx = sqrt(ssx), normy = sqrt(ssy);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] /= normx * normy;                 // A = X0^T Y0 of the unit-norm sets
    // SVD A = U S V^T through the eigen-decomposition of A^T A = V S^2 V^T
    double B[3][3], V[3][3], lam[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { B[r][c] = 0; for (int k = 0; k < 3; ++k) B[r][c] += A[k][r] * A[k][c]; }
    jacobi_eig3(B, V, lam);
    int ord[3] = {0, 1, 2};                                                   // descending singular values
    for (int a = 0; a < 2; ++a)
        for (int b2 = a + 1; b2 < 3; ++b2)
            if (lam[ord[b2]] > lam[ord[a]]) { const int t = ord[a]; ord[a] = ord[b2]; ord[b2] = t; }
    double Vs[3][3], U[3][3], s[3];
    for (int k = 0; k < 3; ++k) {
        s[k] = sqrt(fmax(lam[ord[k]], 0.0));
        for (int r = 0; r < 3; ++r) Vs[r][k] = V[r][ord[k]];
    }
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 3; ++r) {
            double acc = 0;
            for (int c = 0; c < 3; ++c) acc += A[r][c] * Vs[c][k];
            U[r][k] = acc / s[k];
        }
    // third left vector: from A v3 / s3 when well conditioned, else the cross product (rank-2 fits)
    {
        double u3[3];
        for (int r = 0; r < 3; ++r) { double acc = 0; for (int c = 0; c < 3; ++c) acc += A[r][c] * Vs[c][2]; u3[r] = acc; }
        const double cr[3] = {U[1][0] * U[2][1] - U[2][0] * U[1][1], U[2][0] * U[0][1] - U[0][0] * U[2][1],
                              U[0][0] * U[1][1] - U[1][0] * U[0][1]};
        const double dotp = u3[0] * cr[0] + u3[1] * cr[1] + u3[2] * cr[2];
        const double sign = (s[2] > 1e-12 * s[0]) ? (dotp >= 0 ? 1.0 : -1.0) : 1.0;
        for (int r = 0; r < 3; ++r) U[r][2] = sign * cr[r];
    }
    double T[3][3];                                                            // T = V U^T (procrustes.py:64)
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { T[r][c] = 0; for (int k = 0; k < 3; ++k) T[r][c] += Vs[r][k] * U[c][k]; }
    double trace = s[0] + s[1] + s[2];
    if (det3(T) < 0) {                                                         // reflection=False (procrustes.py:66-75)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) T[r][c] -= 2.0 * Vs[r][2] * U[c][2];
        trace -= 2.0 * s[2];
    }
    const double bsc = trace * normx / normy;                                  // scaling=True (procrustes.py:82)
    double cvec[3];
    for (int c = 0; c < 3; ++c) { double acc = 0; for (int k = 0; k < 3; ++k) acc += muy[k] * T[k][c]; cvec[c] = mux[c] - bsc * acc; }
    // aligned = b * pred @ T + c for ALL joints (util3d.py:156-159), then root-relative distance
    double ar[3];
    {
        const int j = nj - 1;
        for (int c = 0; c < 3; ++c) { double acc = 0; for (int k = 0; k < 3; ++k) acc += (double)P[j * 3 + k] * T[k][c]; ar[c] = bsc * acc + cvec[c] - (double)G[j * 3 + c]; }
    }
    for (int j = 0; j < nj; ++j) {
        double ssum = 0;
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += (double)P[j * 3 + k] * T[k][c];
            // the reference returns float32 from the alignment (tfu3d.py:34-38 output_types=tf.float32)
            const double d = ((double)(float)(bsc * acc + cvec[c]) - (double)G[j * 3 + c]) - ((double)(float)(ar[c] + (double)G[(nj - 1) * 3 + c]) - (double)G[(nj - 1) * 3 + c]);
            ssum += d * d;
        }
        dist_pa[(size_t)i * nj + j] = (float)sqrt(ssum);
    }
}

// per joint j: sums over poses of valid, dist, dist_pa, auc score, pck -> out[j][5] (double)
__global__ __launch_bounds__(256) void eval_reduce_kernel(const float* __restrict__ dist, const float* __restrict__ dist_pa,
                                                          const unsigned char* __restrict__ valid, int n, int nj,
                                                          float threshold, double* __restrict__ out) {
    __shared__ double red[256][5];
    const int j = blockIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (!valid[(size_t)i * nj + j]) continue;
        const float d = dist[(size_t)i * nj + j];
        acc[0] += 1.0; acc[1] += d; acc[2] += dist_pa[(size_t)i * nj + j];
        acc[3] += fmaxf(0.f, 1.f - d / threshold);                           // fp32 like the graph (main.py:353-354)
        acc[4] += d <= threshold ? 1.0 : 0.0;
    }
    for (int k = 0; k < 5; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 5; ++k) red[threadIdx.x][k] += red[threadIdx.x + s][k];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 5; ++k) out[j * 5 + k] = red[0][k];
}

int launch_eval_metrics(const float* pred, const float* truth, const unsigned char* valid, int n, int nj,
                        float threshold, float* dist, float* dist_pa, double* sums, hipStream_t stream) {
    hipLaunchKernelGGL(eval_pose_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, pred, truth, valid, n, nj, dist, dist_pa);
    int st = launch_status("eval_pose");
    if (st) return st;
    hipLaunchKernelGGL(eval_reduce_kernel, dim3(nj), dim3(256), 0, stream, dist, dist_pa, valid, n, nj, threshold, sums);
    return launch_status("eval_reduce");
}

}  // namespace metro
