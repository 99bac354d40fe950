// geoac_gridbuild.hip - the evaluation table of the range-dependent sets built ON THE DEVICE.
//
// Replaces, for the grid atmosphere, the set-up work of Set_Slopes_Multi (G2S_MultiDimSpline3D.cpp:306-425,
// G2S_GlobalMultiDimSpline3D.cpp:313-431): per node (i, j) and field the vertical natural splines of f and of the node-centred
// (one-sided at the edges) differences df/dx, df/dy, i.e. 12 tridiagonal systems of nz unknowns per node - 1.2e5 systems of 1400
// unknowns for a 100 x 100 G2S grid - followed by the expansion into the table geoac_rngdep.h evaluates (per (field, kz, node):
// ten cubics F, DxF, DyF, DxyF, Vx, DxVx, DxyVx, Vy, DyVy, DxyVy).  Same arithmetic, in the same order, as the host builder
// geoac_grid_table_eq (geoac_host.cpp), which stays as the checker (tests/test_gpu_gridbuild.py): the slopes are bit-identical
// (FMA contraction is switched off here), the cubic coefficients are formed in double-double where the host uses long double.
//
//   k_gb_columns   one thread per (spline set, node): Thomas sweep down the column (cp/dp kept in a [k][thread] scratch so that
//                  the loads/stores of a wave are contiguous), back substitution, one cubic per vertical segment
//   k_gb_assemble  one thread per (kz, node): the record of 40 (rho: 16) doubles from the cubics of the node and its neighbours
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/geoac_hip.h"

#pragma clang fp contract(off)

namespace {

struct GbParams {
    int nx, ny, nz, glob;
    const double* x; const double* y; const double* z;      // node coordinates
    const double* F;                                        // the field: [nx][ny][nz]
    double* V[3];                                           // cubics of S_f, S_fx, S_fy: [nseg][nn][4]
    double* q[2];                                           // spherical sets: c1 offsets of the Vx, Vy rows (Q12b): [nseg][nn]
    double* cp; double* dp;                                 // Thomas scratch: [nz][n_threads]
    int n_sets;                                             // 3, or 1 for rho (only S_f)
};

// ---- double-double helpers (error-free transformations); used only for the cubic coefficients ----
struct dd { double hi, lo; };
__device__ inline dd two_sum(double a, double b){ double s = a + b, bb = s - a; return { s, (a - (s - bb)) + (b - bb) }; }
__device__ inline dd two_prod(double a, double b){ double p = a * b; return { p, __builtin_fma(a, b, -p) }; }
__device__ inline dd dd_norm(double hi, double lo){ double s = hi + lo; return { s, lo - (s - hi) }; }
__device__ inline dd dd_add(dd a, dd b){ dd s = two_sum(a.hi, b.hi); return dd_norm(s.hi, s.lo + (a.lo + b.lo)); }
__device__ inline dd dd_neg(dd a){ return { -a.hi, -a.lo }; }
__device__ inline dd dd_sub(dd a, dd b){ return dd_add(a, dd_neg(b)); }
__device__ inline dd dd_mul(dd a, dd b){ dd p = two_prod(a.hi, b.hi); return dd_norm(p.hi, p.lo + (a.hi * b.lo + a.lo * b.hi)); }
__device__ inline dd dd_mul_d(dd a, double b){ dd p = two_prod(a.hi, b); return dd_norm(p.hi, p.lo + a.lo * b); }
__device__ inline dd dd_div(dd a, dd b){
    double q1 = a.hi / b.hi;
    dd r = dd_sub(a, dd_mul_d(b, q1));
    double q2 = r.hi / b.hi;
    r = dd_sub(r, dd_mul_d(b, q2));
    double q3 = r.hi / b.hi;
    dd q = two_sum(q1, q2);
    return dd_norm(q.hi, q.lo + q3);
}

// geoac_spline_segment_cubic (geoac_host.cpp; Eval_Spline_f of G2S_Spline1D.cpp:245-281 expanded in powers of t = x - x_k),
// derivative form (c0, c1, 2 c2, 6 c3)
__device__ inline void segment_cubic(double x0, double x1, double f0, double f1, double s0, double s1, double* c){
    const dd h = two_sum(x1, -x0), df = two_sum(f1, -f0);
    const dd A = dd_sub(dd_mul_d(h, s0), df), B = dd_sub(df, dd_mul_d(h, s1));
    const dd h2 = dd_mul(h, h), h3 = dd_mul(h2, h);
    const dd c2 = dd_div(dd_sub(B, dd_mul_d(A, 2.0)), h2), c3 = dd_div(dd_sub(A, B), h3);
    c[0] = f0;
    c[1] = s0;
    c[2] = 2.0 * c2.hi + 2.0 * c2.lo;
    c[3] = 6.0 * c3.hi + 6.0 * c3.lo;
}

__global__ void __launch_bounds__(256) k_gb_columns(GbParams P){
    const int nn = P.nx * P.ny, nseg = P.nz - 1;
    const int T = P.n_sets * nn;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if(tid >= T) return;
    const int s = tid / nn, node = tid - s * nn;
    const int i = node / P.ny, j = node - i * P.ny;
    const int n = P.nz;
    // the column this spline runs through: node values, or the centred / one-sided difference along the first / second axis
    const double* c0 = P.F + (size_t)node * n;
    const double* cu = c0; const double* cd = c0;
    double dxy = 1.0;
    if(s == 1){
        const int iu = (i + 1 < P.nx) ? i + 1 : P.nx - 1, id = (i > 0) ? i - 1 : 0;
        cu = P.F + ((size_t)iu * P.ny + j) * n; cd = P.F + ((size_t)id * P.ny + j) * n;
        dxy = P.x[iu] - P.x[id];
    } else if(s == 2){
        const int ju = (j + 1 < P.ny) ? j + 1 : P.ny - 1, jd = (j > 0) ? j - 1 : 0;
        cu = P.F + ((size_t)i * P.ny + ju) * n; cd = P.F + ((size_t)i * P.ny + jd) * n;
        dxy = P.y[ju] - P.y[jd];
    }
    auto col = [&](int k) -> double { return (s == 0) ? c0[k] : (cu[k] - cd[k]) / dxy; };
    const bool quirk = (P.glob != 0) && (s > 0);                  // Q12a: interior right-hand sides use (d[i] - d[i+1])
    const double* x = P.z;
    double* cp = P.cp + tid; double* dp = P.dp + tid;
    const size_t st = (size_t)T;

    // ---- forward sweep (geoac_natural_spline_slopes / slopes_q12a, geoac_host.cpp) ----
    double fm = col(0), fi = col(1);
    double h0 = x[1] - x[0];
    double b = 2.0 / h0, c = 1.0 / h0, d = 3.0 * (fi - fm) / (h0 * h0);
    double cpp = c / b, dpp = d / b;
    cp[0] = cpp; dp[0] = dpp;
    for(int k = 1; k < n - 1; k++){
        const double fp = col(k + 1);
        const double hl = x[k] - x[k - 1], hr = x[k + 1] - x[k];
        const double a = 1.0 / hl;
        b = 2.0 * (1.0 / hl + 1.0 / hr);
        c = 1.0 / hr;
        d = quirk ? 3.0 * ((fi - fp) / (hl * hl) + (fp - fi) / (hr * hr))
                  : 3.0 * ((fi - fm) / (hl * hl) + (fp - fi) / (hr * hr));
        const double den = b - cpp * a;
        cpp = c / den;
        dpp = (d - dpp * a) / den;
        cp[(size_t)k * st] = cpp; dp[(size_t)k * st] = dpp;
        fm = fi; fi = fp;
    }
    {
        const double hn = x[n - 1] - x[n - 2];
        const double a = 1.0 / hn;
        b = 2.0 / hn;
        d = 3.0 * (fi - fm) / (hn * hn);
        dpp = (d - dpp * a) / (b - cpp * a);
    }
    // ---- back substitution, one cubic per segment on the way up ----
    double s_hi = dpp, f_hi = fi;                                  // slope and value at node k + 1
    double* V = P.V[s];
    for(int k = n - 2; k >= 0; k--){
        const double s_lo = dp[(size_t)k * st] - cp[(size_t)k * st] * s_hi;
        const double f_lo = col(k);
        double cc[4];
        segment_cubic(x[k], x[k + 1], f_lo, f_hi, s_lo, s_hi, cc);
        double* o = V + (((size_t)k * nn) + node) * 4;
        o[0] = cc[0]; o[1] = cc[1]; o[2] = cc[2]; o[3] = cc[3];
        if(quirk) P.q[s - 1][(size_t)k * nn + node] = (f_hi - f_lo) / (x[k + 1] - x[k]);
        s_hi = s_lo; f_hi = f_lo;
    }
    (void)nseg;
}

__global__ void __launch_bounds__(256) k_gb_assemble(GbParams P, double* out, int stride){
    const int nn = P.nx * P.ny, nseg = P.nz - 1;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if(tid >= (long long)nseg * nn) return;
    const int k = (int)(tid / nn), node = (int)(tid - (long long)k * nn);
    const int i = node / P.ny, j = node - i * P.ny;
    const int iu = (i + 1 < P.nx) ? i + 1 : P.nx - 1, id = (i > 0) ? i - 1 : 0;
    const int ju = (j + 1 < P.ny) ? j + 1 : P.ny - 1, jd = (j > 0) ? j - 1 : 0;
    const double ix = 1.0 / (P.x[iu] - P.x[id]), iy = 1.0 / (P.y[ju] - P.y[jd]);
    auto at = [&](const double* V, int a, int b2) -> const double* { return V + (((size_t)k * nn) + (size_t)a * P.ny + b2) * 4; };
    double* r = out + (((size_t)k * nn) + node) * stride;
    #pragma unroll
    for(int c = 0; c < 4; c++){
        const double* V0 = P.V[0];
        r[0 + c]  = at(V0, i, j)[c];
        r[4 + c]  = (at(V0, iu, j)[c] - at(V0, id, j)[c]) * ix;
        r[8 + c]  = (at(V0, i, ju)[c] - at(V0, i, jd)[c]) * iy;
        r[12 + c] = (at(V0, iu, ju)[c] - at(V0, iu, jd)[c] - at(V0, id, ju)[c] + at(V0, id, jd)[c]) * (ix * iy);
        if(stride == 40){
            const double* Vx = P.V[1]; const double* Vy = P.V[2];
            r[16 + c] = at(Vx, i, j)[c];
            r[20 + c] = (at(Vx, iu, j)[c] - at(Vx, id, j)[c]) * ix;
            r[24 + c] = (at(Vx, iu, ju)[c] - at(Vx, iu, jd)[c] - at(Vx, id, ju)[c] + at(Vx, id, jd)[c]) * (ix * iy);
            r[28 + c] = at(Vy, i, j)[c];
            r[32 + c] = (at(Vy, i, ju)[c] - at(Vy, i, jd)[c]) * iy;
            r[36 + c] = (at(Vy, iu, ju)[c] - at(Vy, iu, jd)[c] - at(Vy, id, ju)[c] + at(Vy, id, jd)[c]) * (ix * iy);
        }
    }
    if(stride == 40 && P.glob){
        r[16 + 1] -= P.q[0][(size_t)k * nn + node];
        r[28 + 1] -= P.q[1][(size_t)k * nn + node];
    }
}

// Cartesian set: V_x = S_fx and V_y = S_fy are splines of centred differences of the node values, D_x F and D_y F are the same
// differences of the spline coefficients - the same cubics up to rounding (the spline systems are linear and share the z grid; measured
// < 1e-10 of the coefficient scale).  The kernels read the packed form: eight cubics  F, DxF, DyF, DxyF | DxVx, DxyVx, DyVy, DxyVy  per
// (field, kz, node), 256 bytes = two cache lines, line-aligned.  One thread per 16-byte chunk.
__global__ void k_gb_pack8(const double* __restrict__ t40, double* __restrict__ t8, long long n_rec){
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n_rec * 16) return;
    const long long rec = i >> 4; const int ch = (int)(i & 15);
    const int cub = ch >> 1;
    const int src = cub < 4 ? cub : (cub < 6 ? cub + 1 : cub + 2);         // 4, 5 <- DxVx, DxyVx (5, 6);  6, 7 <- DyVy, DxyVy (8, 9)
    const double2 v = *(const double2*)(t40 + rec * 40 + 4 * src + 2 * (ch & 1));
    *(double2*)(t8 + rec * 32 + 2 * ch) = v;
}

}  // namespace

// packed Cartesian table (3 n_seg n_node records of 32 doubles, then the rho block unchanged) from the full one
extern "C" size_t geoac_gridpack_doubles(int nx, int ny, int nz){
    const size_t nn = (size_t)nx * ny, nseg = (size_t)(nz - 1);
    return 3 * nseg * nn * 32 + nseg * nn * 16;
}
extern "C" hipError_t geoac_gridpack_launch(int nx, int ny, int nz, const double* d_tab, double* d_tab8, hipStream_t s){
    const size_t nn = (size_t)nx * ny, nseg = (size_t)(nz - 1);
    const long long n_rec = (long long)(3 * nseg * nn);
    hipLaunchKernelGGL(k_gb_pack8, dim3((unsigned)((n_rec * 16 + 255) / 256)), dim3(256), 0, s, d_tab, d_tab8, n_rec);
    hipError_t e = hipGetLastError();
    if(e != hipSuccess) return e;
    return hipMemcpyAsync(d_tab8 + (size_t)n_rec * 32, d_tab + (size_t)n_rec * 40, sizeof(double) * nseg * nn * 16, hipMemcpyDeviceToDevice, s);
}

// Builds the whole table (geoac_grid_table_size doubles at d_tab) from the device copies of the node coordinates and of the four
// fields ([4][nx][ny][nz]: T, u, v, rho).  d_work: scratch of geoac_gridbuild_work_doubles(nx, ny, nz) doubles.
extern "C" size_t geoac_gridbuild_work_doubles(int nx, int ny, int nz){
    const size_t nn = (size_t)nx * ny, nseg = (size_t)(nz - 1);
    return 3 * nseg * nn * 4 + 2 * nseg * nn + 2 * (size_t)nz * 3 * nn;
}

extern "C" hipError_t geoac_gridbuild_launch(int glob, int nx, int ny, int nz, const double* d_x, const double* d_y, const double* d_z,
                                             const double* d_fields, double* d_work, double* d_tab, hipStream_t s){
    const size_t nn = (size_t)nx * ny, nseg = (size_t)(nz - 1);
    GbParams P;
    P.nx = nx; P.ny = ny; P.nz = nz; P.glob = glob;
    P.x = d_x; P.y = d_y; P.z = d_z;
    double* w = d_work;
    for(int q = 0; q < 3; q++){ P.V[q] = w; w += nseg * nn * 4; }
    for(int q = 0; q < 2; q++){ P.q[q] = w; w += nseg * nn; }
    P.cp = w; w += (size_t)nz * 3 * nn;
    P.dp = w;
    for(int f = 0; f < 4; f++){
        P.F = d_fields + (size_t)f * nn * nz;
        P.n_sets = (f == 3) ? 1 : 3;                               // rho is only ever evaluated through Eval_Spline_f
        const int T = P.n_sets * (int)nn;
        hipLaunchKernelGGL(k_gb_columns, dim3((T + 255) / 256), dim3(256), 0, s, P);
        const long long total = (long long)nseg * (long long)nn;
        double* out = (f < 3) ? d_tab + (size_t)f * nseg * nn * 40 : d_tab + (size_t)3 * nseg * nn * 40;
        hipLaunchKernelGGL(k_gb_assemble, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P, out, (f < 3) ? 40 : 16);
    }
    return hipGetLastError();
}
