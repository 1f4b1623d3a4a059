// Alternative decode heads of the reference: the step AFTER the soft-argmax.
//   * backproject_kernel: `bone-lengths` / `bone-lengths-true` / `true-root-depth` scale recovery
//     (reference src/model/volumetric.py:171-199): image coordinates (heatmap_to_image :288-295), camera rays
//     (matmul_joint_coords with inv_intrinsics :221-222), delta_z, the per-pose z-offset solve
//     (src/model/bone_length_based_backproj.py:38-62) and back_project (:284-285), optional root-relative
//     (tfu3d.py:23-25) and export permutation (main.py:119-127).
//   * to_orig_cam_kernel: rotation to the original camera with joints mirrored when det(R) <= 0
//     (volumetric.py:277-281).
// The z-offset solve of the reference is scipy.optimize.least_squares(method='lm') = MINPACK lmder with a Jacobian
// that is NOT the derivative of the residual ((z*c+d)/len instead of (z*c+d/2)/len, :55-56): where it stops
// depends on MINPACK's step-acceptance history, so `lmder1` below restates lmder / qrfac / lmpar / qrsolv for one
// unknown in fp64, operation by operation (mode 2, diag = 1, ftol = xtol = gtol = 1e-8, factor = 100,
// maxfev = 100; scipy/optimize/_lsq/least_squares.py call_minpack).  The fp32 part mirrors NumPy on the fp32
// tensors TF hands to the py_func: no FMA contraction anywhere in this file.
// One thread per pose (E, J <= 64): the work is a few hundred flops per pose, the point is parity, not speed.
#include "metro_common.h"

#pragma clang fp contract(off)

namespace metro {

constexpr int HEAD_MAX = 64;

struct BackprojectArgs {
    const float* coords01;      // [n][nj][3]  soft-argmax output in [0,1], head order, (x,y,z)
    const float* inv_k;         // [n][9]
    const double* targets;      // [ne] or [n][ne]   (mode 0)
    const float* root_z;        // [n]               (mode 1: true-root-depth)
    const int* edges;           // [ne][2] head joint indices
    float* out;                 // [n][n_out][3]
    float* z_out;               // [n] or null
    int n, nj, ne, per_pose_targets, mode;
    float lrc, half_off, box;
    int root_relative, n_out;
    int perm[HEAD_MAX];
};

struct LmProblem {
    const double* c; const double* d; const double* e; const double* t; int m;
};

__device__ inline void lm_fn(const LmProblem& p, double z, double* f) {
    for (int i = 0; i < p.m; ++i) f[i] = sqrt(z * z * p.c[i] + z * p.d[i] + p.e[i]) - p.t[i];
}
__device__ inline void lm_jac(const LmProblem& p, double z, double* j) {
    for (int i = 0; i < p.m; ++i) j[i] = (z * p.c[i] + p.d[i]) / sqrt(z * z * p.c[i] + z * p.d[i] + p.e[i]);
}
__device__ inline double lm_enorm(const double* v, int m) {
    double s = 0.0;
    for (int i = 0; i < m; ++i) s += v[i] * v[i];
    return sqrt(s);
}

// MINPACK lmder, n = 1, mode = 2, diag = 1 (see oracle/lm1.py for the same sequence in Python)
__device__ double lmder1(const LmProblem& prob, double x0) {
    const double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8, factor = 100.0, diag = 1.0;
    const double epsmch = 2.220446049250313e-16, dwarf = 2.2250738585072014e-308;
    const int maxfev = 100, m = prob.m;
    double fvec[HEAD_MAX], f2[HEAD_MAX], fjac[HEAD_MAX], wa4[HEAD_MAX];
    double x = x0;
    lm_fn(prob, x, fvec);
    int nfev = 1, it = 1, info = 0;
    double fnorm = lm_enorm(fvec, m);
    double par = 0.0, delta = 0.0, xnorm = 0.0;
    while (true) {
        lm_jac(prob, x, fjac);
        const double acnorm = lm_enorm(fjac, m);          // qrfac
        double ajnorm = acnorm;
        if (ajnorm != 0.0) {
            if (fjac[0] < 0.0) ajnorm = -ajnorm;
            for (int i = 0; i < m; ++i) fjac[i] = fjac[i] / ajnorm;
            fjac[0] += 1.0;
        }
        const double r = -ajnorm;
        if (it == 1) {
            xnorm = sqrt((diag * x) * (diag * x));
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        for (int i = 0; i < m; ++i) wa4[i] = fvec[i];
        if (fjac[0] != 0.0) {
            double s = 0.0;
            for (int i = 0; i < m; ++i) s += fjac[i] * wa4[i];
            const double temp = -s / fjac[0];
            for (int i = 0; i < m; ++i) wa4[i] += fjac[i] * temp;
        }
        const double qtf = wa4[0];
        double gnorm = 0.0;
        if (fnorm != 0.0 && acnorm != 0.0) {
            const double s = r * (qtf / fnorm);
            gnorm = fmax(gnorm, fabs(s / acnorm));
        }
        if (gnorm <= gtol) { info = 4; break; }
        while (true) {
            // ---- lmpar ----
            double p = r == 0.0 ? 0.0 : qtf / r;
            int liter = 0;
            double wa2 = diag * p;
            double dxnorm = sqrt(wa2 * wa2);
            double fp = dxnorm - delta;
            double par_out;
            if (fp <= 0.1 * delta) {
                par_out = 0.0;
            } else {
                double parl = 0.0;
                if (r != 0.0) {
                    double w = diag * (wa2 / dxnorm);
                    w = w / r;
                    const double temp = sqrt(w * w);
                    parl = ((fp / delta) / temp) / temp;
                }
                const double gw = (r * qtf) / diag;
                const double gn = sqrt(gw * gw);
                double paru = gn / delta;
                if (paru == 0.0) paru = dwarf / fmin(delta, 0.1);
                double pl = fmax(par, parl);
                pl = fmin(pl, paru);
                if (pl == 0.0) pl = gn / dxnorm;
                double sdiag = 0.0;
                while (true) {
                    ++liter;
                    if (pl == 0.0) pl = fmax(dwarf, 0.001 * paru);
                    const double sd = sqrt(pl) * diag;
                    double rr = r, wa = qtf, qtbpj = 0.0;          // qrsolv
                    if (sd != 0.0) {
                        double sn, cs;
                        if (fabs(rr) < fabs(sd)) {
                            const double cotan = rr / sd;
                            sn = 0.5 / sqrt(0.25 + 0.25 * cotan * cotan);
                            cs = sn * cotan;
                        } else {
                            const double tn = sd / rr;
                            cs = 0.5 / sqrt(0.25 + 0.25 * tn * tn);
                            sn = cs * tn;
                        }
                        rr = cs * rr + sn * sd;
                        const double t2 = cs * wa + sn * qtbpj;
                        qtbpj = -sn * wa + cs * qtbpj;
                        wa = t2;
                    }
                    sdiag = rr;
                    p = sdiag != 0.0 ? wa / sdiag : 0.0;
                    wa2 = diag * p;
                    dxnorm = sqrt(wa2 * wa2);
                    const double temp = fp;
                    fp = dxnorm - delta;
                    if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || liter == 10) break;
                    double w = diag * (wa2 / dxnorm);
                    w = w / sdiag;
                    const double tw = sqrt(w * w);
                    const double parc = ((fp / delta) / tw) / tw;
                    if (fp > 0.0) parl = fmax(parl, pl);
                    if (fp < 0.0) paru = fmin(paru, pl);
                    pl = fmax(parl, pl + parc);
                }
                par_out = liter == 0 ? 0.0 : pl;
            }
            par = par_out;
            // ---- back in lmder ----
            const double wa1 = -p;
            const double x2 = x + wa1;
            const double wa3 = diag * wa1;
            const double pnorm = sqrt(wa3 * wa3);
            if (it == 1) delta = fmin(delta, pnorm);
            lm_fn(prob, x2, f2);
            ++nfev;
            const double fnorm1 = lm_enorm(f2, m);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) { const double q = fnorm1 / fnorm; actred = 1.0 - q * q; }
            const double w3 = r * wa1;
            const double temp1 = sqrt(w3 * w3) / fnorm;
            const double temp2 = (sqrt(par) * pnorm) / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            double ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp;
                if (actred >= 0.0) temp = 0.5;
                else temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * fmin(delta, pnorm / 0.1);
                par = par / temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
                x = x2;
                for (int i = 0; i < m; ++i) fvec[i] = f2[i];
                xnorm = sqrt((diag * x) * (diag * x));
                fnorm = fnorm1;
                ++it;
            }
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0 && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= epsmch * xnorm) info = 7;
            if (gnorm <= epsmch) info = 8;
            if (info != 0) break;
            if (ratio >= 1e-4) break;
        }
        if (info != 0) break;
    }
    return x;
}

__global__ __launch_bounds__(64) void backproject_kernel(BackprojectArgs a) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= a.n) return;
    const float* c01 = a.coords01 + (size_t)img * a.nj * 3;
    const float* k = a.inv_k + (size_t)img * 9;
    float cam[HEAD_MAX][3], dz[HEAD_MAX];
    const float zroot = c01[(a.nj - 1) * 3 + 2];
    for (int j = 0; j < a.nj; ++j) {
        // heatmap_to_image (volumetric.py:288-295): coords * last_receptive_center (+ stride // 2)
        float u = c01[j * 3 + 0] * a.lrc, v = c01[j * 3 + 1] * a.lrc;
        u = u + a.half_off; v = v + a.half_off;
        for (int i = 0; i < 3; ++i) cam[j][i] = (k[i * 3 + 0] * u + k[i * 3 + 1] * v) + k[i * 3 + 2] * 1.0f;
        dz[j] = (c01[j * 3 + 2] - zroot) * a.box;
    }
    float z_off;
    if (a.mode == 1) {
        z_off = a.root_z[img];
    } else {
        double c[HEAD_MAX], d[HEAD_MAX], e[HEAD_MAX];
        for (int q = 0; q < a.ne; ++q) {
            const int i = a.edges[q * 2], j = a.edges[q * 2 + 1];
            float av[3], bv[3];
            for (int t = 0; t < 3; ++t) {
                av[t] = cam[i][t] - cam[j][t];
                bv[t] = cam[i][t] * dz[i] - cam[j][t] * dz[j];
            }
            // np.sum over 3 fp32 elements: sequential
            const float cf = (av[0] * av[0] + av[1] * av[1]) + av[2] * av[2];
            const float df = ((2.0f * av[0]) * bv[0] + (2.0f * av[1]) * bv[1]) + (2.0f * av[2]) * bv[2];
            const float ef = (bv[0] * bv[0] + bv[1] * bv[1]) + bv[2] * bv[2];
            c[q] = (double)cf; d[q] = (double)df; e[q] = (double)ef;
        }
        LmProblem prob;
        prob.c = c; prob.d = d; prob.e = e; prob.m = a.ne;
        prob.t = a.targets + (a.per_pose_targets ? (size_t)img * a.ne : 0);
        z_off = (float)lmder1(prob, 2000.0);                  // initial_guess=2000, np.float32 result
    }
    if (a.z_out) a.z_out[img] = z_off;
    // back_project (volumetric.py:284-285), then optional root_relative + export gather
    float root[3];
    for (int t = 0; t < 3; ++t) root[t] = a.root_relative ? cam[a.nj - 1][t] * (dz[a.nj - 1] + z_off) : 0.0f;
    float* o = a.out + (size_t)img * a.n_out * 3;
    for (int r = 0; r < a.n_out; ++r) {
        const int j = a.perm[r];
        const float s = dz[j] + z_off;
        for (int t = 0; t < 3; ++t) o[r * 3 + t] = cam[j][t] * s - root[t];
    }
}

__global__ __launch_bounds__(64) void to_orig_cam_kernel(const float* __restrict__ x, const float* __restrict__ rot,
                                                         const int* __restrict__ mirror, float* __restrict__ out,
                                                         int n, int nj) {
    const int img = blockIdx.x;
    const int j = threadIdx.x;
    if (img >= n || j >= nj) return;
    const float* r = rot + (size_t)img * 9;
    // tf.linalg.det in fp32 on a 3x3; the sign is what matters (volumetric.py:279-281): evaluate it in fp64
    const double det = (double)r[0] * ((double)r[4] * r[8] - (double)r[5] * r[7]) -
                       (double)r[1] * ((double)r[3] * r[8] - (double)r[5] * r[6]) +
                       (double)r[2] * ((double)r[3] * r[7] - (double)r[4] * r[6]);
    const int src = det > 0.0 ? j : mirror[j];
    const float* p = x + ((size_t)img * nj + src) * 3;
    float* o = out + ((size_t)img * nj + j) * 3;
    for (int i = 0; i < 3; ++i) o[i] = (r[i * 3 + 0] * p[0] + r[i * 3 + 1] * p[1]) + r[i * 3 + 2] * p[2];
}

int launch_backproject(const float* coords01, const float* inv_k, const double* targets, int per_pose_targets,
                       const float* root_z, const int* edges, int n, int nj, int ne, const MetroSpec& spec,
                       int root_relative, int permute, float* out, float* z_out, hipStream_t stream) {
    BackprojectArgs a;
    a.coords01 = coords01; a.inv_k = inv_k; a.targets = targets; a.root_z = root_z; a.edges = edges;
    a.out = out; a.z_out = z_out;
    a.n = n; a.nj = nj; a.ne = ne; a.per_pose_targets = per_pose_targets; a.mode = root_z != nullptr ? 1 : 0;
    const int last = spec.proc_side - 1;
    a.lrc = (float)(last - (last % spec.stride) - 1);
    a.half_off = spec.centered_stride ? (float)(spec.stride / 2) : 0.0f;
    a.box = spec.box_size_mm;
    a.root_relative = root_relative;
    a.n_out = permute ? spec.n_joints_out : nj;
    for (int i = 0; i < HEAD_MAX; ++i) a.perm[i] = permute ? (i < spec.n_joints_out ? spec.permutation[i] : 0) : i;
    hipLaunchKernelGGL(backproject_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, a);
    return launch_status("backproject");
}

// heatmap_to_25d (volumetric.py:298-300): image-pixel x, y and z * box_size per head joint
__global__ __launch_bounds__(256) void heatmap_to_25d_kernel(const float* __restrict__ c01, float* __restrict__ out, int total,
                                                             float lrc, float half_off, float box) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // one (pose, joint)
    if (i >= total) return;
    float u = c01[i * 3 + 0] * lrc, v = c01[i * 3 + 1] * lrc;
    out[i * 3 + 0] = u + half_off;
    out[i * 3 + 1] = v + half_off;
    out[i * 3 + 2] = c01[i * 3 + 2] * box;
}

int launch_heatmap_to_25d(const float* coords01, float* out, int n, const MetroSpec& spec, hipStream_t stream) {
    const int last = spec.proc_side - 1;
    const int total = n * spec.n_joints_head;
    hipLaunchKernelGGL(heatmap_to_25d_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, coords01, out, total,
                       (float)(last - (last % spec.stride) - 1), spec.centered_stride ? (float)(spec.stride / 2) : 0.0f,
                       spec.box_size_mm);
    return launch_status("heatmap_to_25d");
}

int launch_to_orig_cam(const float* x, const float* rot, const int* mirror, float* out, int n, int nj, hipStream_t stream) {
    hipLaunchKernelGGL(to_orig_cam_kernel, dim3(n), dim3(64), 0, stream, x, rot, mirror, out, n, nj);
    return launch_status("to_orig_cam");
}

}  // namespace metro
