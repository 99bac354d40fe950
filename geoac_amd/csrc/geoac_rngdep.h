// geoac_rngdep.h - device code of the range-dependent Cartesian set (GeoAc3D.RngDep): included by geoac_kernels.hip.
//
// Atmosphere = grid of vertical profiles.  The reference (Code/Atmo/G2S_MultiDimSpline3D.cpp) keeps, per node (i, j), natural
// cubic splines in z of f, of the centred-difference df/dx and of df/dy, and evaluates a field at (x, y, z) as a bicubic
// Hermite patch over the cell whose corner data (f, f_x, f_y, f_xy) are vertical-spline values and horizontal finite
// differences of vertical-spline values; every derivative the equation set needs is again its own bicubic patch of
// differentiated data (Eval_Spline_AllOrder1/2, :1156-1593).  Here:
//   * table  gtab[field][kz][ix][iy][12] : per node and vertical segment the three cubics (f, df/dx, df/dy) in the
//     derivative-friendly form (c0, c1, 2c2, 6c3); all 16 nodes of a 4x4 neighbourhood at one kz are within 96*nx*ny bytes;
//   * one pass per vertical quantity over the 4x4 neighbourhood -> corner values and finite differences,
//   * the 16x16 matrix product + power sums of the reference are evaluated as the equivalent tensor Hermite form.
// Quirk Q11 kept: the scalar evaluators and the d2f/dz2 patch scale the y-derivative rows by the x cell size.
#ifndef GEOAC_RNGDEP_H_
#define GEOAC_RNGDEP_H_

struct GridLoc {
    int kz;                 // vertical segment
    int nb[4][4];           // node index (ix*ny + iy) of the clamped 4x4 neighbourhood, [p][q], p <-> x, q <-> y
    double t;               // z - z0[kz]
    double xs, ys;          // position inside the cell, scaled to [0, 1]
    double dxs, dys;        // cell sizes (dx_scalar, dy_scalar)
    double idx[2], idy[2];  // 1 / (X[a+2] - X[a]), 1 / (Y[b+2] - Y[b]) : spans of the finite differences at the cell corners
};

// locate (x, y, z) (already clamped to the grid): cell, neighbourhood, vertical segment.  kz_hint < 0: search from scratch.
DEVINL void grid_locate(const GeoacDevParams& P, double x, double y, double z, int kz_hint, GridLoc& L){
    const int nx = P.gnx, ny = P.gny;
    int kx = 0, ky = 0;
    for(int i = 1; i < nx - 1; i++) kx += (x >= P.gx[i]) ? 1 : 0;       // last i with x >= X[i], capped at nx-2 (5-ish nodes: branch-free scan)
    for(int j = 1; j < ny - 1; j++) ky += (y >= P.gy[j]) ? 1 : 0;
    int kz;
    if(kz_hint < 0){
        double span = P.x_max - P.x_min;                                  // x_min/x_max hold the z range for this set
        kz = (int)((z - P.x_min) / span * (double)P.nseg);
        kz = kz < 0 ? 0 : (kz > P.nseg - 1 ? P.nseg - 1 : kz);
    } else kz = kz_hint;
    while(kz > 0 && z < P.gz[kz]) kz--;
    while(kz < P.nseg - 1 && z > P.gz[kz + 1]) kz++;
    L.kz = kz;
    L.t = z - P.gz[kz];
    int ix[4] = { kx > 0 ? kx - 1 : 0, kx, kx + 1, kx + 2 < nx ? kx + 2 : nx - 1 };
    int iy[4] = { ky > 0 ? ky - 1 : 0, ky, ky + 1, ky + 2 < ny ? ky + 2 : ny - 1 };
    #pragma unroll
    for(int p = 0; p < 4; p++){
        #pragma unroll
        for(int q = 0; q < 4; q++) L.nb[p][q] = ix[p] * ny + iy[q];
    }
    const double X1 = P.gx[kx], X2 = P.gx[kx + 1], Y1 = P.gy[ky], Y2 = P.gy[ky + 1];
    L.dxs = X2 - X1; L.dys = Y2 - Y1;
    L.xs = (x - X1) / L.dxs; L.ys = (y - Y1) / L.dys;
    L.idx[0] = 1.0 / (X2 - P.gx[ix[0]]); L.idx[1] = 1.0 / (P.gx[ix[3]] - X1);
    L.idy[0] = 1.0 / (Y2 - P.gy[iy[0]]); L.idy[1] = 1.0 / (P.gy[iy[3]] - Y1);
}

// Hermite basis on [0,1] and its derivative: value weights h0, h1 and slope weights g0, g1 of the two cell edges
struct Herm { double h0, h1, g0, g1, dh0, dh1, dg0, dg1; };
DEVINL Herm hermite(double s){
    Herm H; double s2 = s * s, s3 = s2 * s;
    H.h1 = 3.0 * s2 - 2.0 * s3; H.h0 = 1.0 - H.h1;
    H.g0 = s - 2.0 * s2 + s3;   H.g1 = s3 - s2;
    H.dh1 = 6.0 * (s - s2);     H.dh0 = -H.dh1;
    H.dg0 = 1.0 - 4.0 * s + 3.0 * s2; H.dg1 = 3.0 * s2 - 2.0 * s;
    return H;
}

// corner data of one bicubic patch: F, FX (already times dx), FY (times dy), FXY (times dx dy), [a][b] = corner (x edge a, y edge b)
struct Patch { double F[2][2], FX[2][2], FY[2][2], FXY[2][2]; };

// value and (optionally) the two first derivatives in SCALED coordinates of the patch
template <bool WANT_DX, bool WANT_DY>
DEVINL void patch_eval(const Patch& B, const Herm& hx, const Herm& hy, double& val, double& dsx, double& dsy){
    double U[2], W[2], Ud[2], Wd[2];
    #pragma unroll
    for(int a = 0; a < 2; a++){
        U[a] = B.F[a][0] * hy.h0 + B.F[a][1] * hy.h1 + B.FY[a][0] * hy.g0 + B.FY[a][1] * hy.g1;
        W[a] = B.FX[a][0] * hy.h0 + B.FX[a][1] * hy.h1 + B.FXY[a][0] * hy.g0 + B.FXY[a][1] * hy.g1;
        if(WANT_DY){
            Ud[a] = B.F[a][0] * hy.dh0 + B.F[a][1] * hy.dh1 + B.FY[a][0] * hy.dg0 + B.FY[a][1] * hy.dg1;
            Wd[a] = B.FX[a][0] * hy.dh0 + B.FX[a][1] * hy.dh1 + B.FXY[a][0] * hy.dg0 + B.FXY[a][1] * hy.dg1;
        }
    }
    val = U[0] * hx.h0 + U[1] * hx.h1 + W[0] * hx.g0 + W[1] * hx.g1;
    if(WANT_DX) dsx = U[0] * hx.dh0 + U[1] * hx.dh1 + W[0] * hx.dg0 + W[1] * hx.dg1;
    if(WANT_DY) dsy = Ud[0] * hx.h0 + Ud[1] * hx.h1 + Wd[0] * hx.g0 + Wd[1] * hx.g1;
}

// node record of field f at (kz, node): 12 doubles {S_f, S_fx, S_fy} x (c0, c1, 2c2, 6c3)
DEVINL const double* grid_rec(const GeoacDevParams& P, int field, int kz, int node){
    return P.gtab + (((size_t)field * P.nseg + kz) * (size_t)(P.gnx * P.gny) + node) * 12;
}
DEVINL double cubic_val(const double* c, double t, double t6){     // c0 + t (c1 + t/6 (3 d2 + e3 t))
    return __builtin_fma(t, __builtin_fma(t6, __builtin_fma(t, c[3], 3.0 * c[2]), c[1]), c[0]);
}
DEVINL double cubic_d1(const double* c, double t, double th){      // c1 + t/2 (2 d2 + e3 t)
    return __builtin_fma(th, __builtin_fma(t, c[3], 2.0 * c[2]), c[1]);
}
DEVINL double cubic_d2(const double* c, double t){ return __builtin_fma(t, c[3], c[2]); }

// finite differences at the four cell corners from values g[p][q] on the 4x4 neighbourhood (BiCubic_Deriv_*, :568-800)
DEVINL void fd_x(const double g[4][4], const GridLoc& L, double o[2][2]){
    #pragma unroll
    for(int a = 0; a < 2; a++){
        #pragma unroll
        for(int b = 0; b < 2; b++) o[a][b] = (g[a + 2][b + 1] - g[a][b + 1]) * L.idx[a];
    }
}
DEVINL void fd_y(const double g[4][4], const GridLoc& L, double o[2][2]){
    #pragma unroll
    for(int a = 0; a < 2; a++){
        #pragma unroll
        for(int b = 0; b < 2; b++) o[a][b] = (g[a + 1][b + 2] - g[a + 1][b]) * L.idy[b];
    }
}
DEVINL void fd_xy(const double g[4][4], const GridLoc& L, double o[2][2]){
    #pragma unroll
    for(int a = 0; a < 2; a++){
        #pragma unroll
        for(int b = 0; b < 2; b++) o[a][b] = (g[a + 2][b + 2] - g[a + 2][b] - g[a][b + 2] + g[a][b]) * (L.idx[a] * L.idy[b]);
    }
}
DEVINL void corners(const double g[4][4], double o[2][2]){
    #pragma unroll
    for(int a = 0; a < 2; a++){
        #pragma unroll
        for(int b = 0; b < 2; b++) o[a][b] = g[a + 1][b + 1];
    }
}
DEVINL void scale22(double o[2][2], double s){
    #pragma unroll
    for(int a = 0; a < 2; a++){
        #pragma unroll
        for(int b = 0; b < 2; b++) o[a][b] *= s;
    }
}
DEVINL void copy22(const double i[2][2], double o[2][2], double s){
    #pragma unroll
    for(int a = 0; a < 2; a++){
        #pragma unroll
        for(int b = 0; b < 2; b++) o[a][b] = i[a][b] * s;
    }
}

// which of the three vertical cubics of a node record, and which z-derivative of it
template <int WHICH, int ZDER>
DEVINL void nbhd(const GeoacDevParams& P, int field, const GridLoc& L, double g[4][4]){
    const double t = L.t, th = 0.5 * t, t6 = t * (1.0 / 6.0);
    #pragma unroll
    for(int p = 0; p < 4; p++){
        #pragma unroll
        for(int q = 0; q < 4; q++){
            const double* c = grid_rec(P, field, L.kz, L.nb[p][q]) + 4 * WHICH;
            g[p][q] = (ZDER == 0) ? cubic_val(c, t, t6) : (ZDER == 1) ? cubic_d1(c, t, th) : cubic_d2(c, t);
        }
    }
}
template <int WHICH, int ZDER>
DEVINL void nbhd_corners(const GeoacDevParams& P, int field, const GridLoc& L, double o[2][2]){
    const double t = L.t, th = 0.5 * t, t6 = t * (1.0 / 6.0);
    #pragma unroll
    for(int a = 0; a < 2; a++){
        #pragma unroll
        for(int b = 0; b < 2; b++){
            const double* c = grid_rec(P, field, L.kz, L.nb[a + 1][b + 1]) + 4 * WHICH;
            o[a][b] = (ZDER == 0) ? cubic_val(c, t, t6) : (ZDER == 1) ? cubic_d1(c, t, th) : cubic_d2(c, t);
        }
    }
}

// Eval_Spline_AllOrder1 (ORDER2 = false, :1156-1339) / AllOrder2 (:1341-1593):
// out = f, f_x, f_y, f_z [, f_xx, f_yy, f_zz, f_xy, f_xz, f_yz]
template <bool ORDER2>
DEVINL void grid_eval_all(const GeoacDevParams& P, int field, const GridLoc& L, double* out){
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    const double dxs = L.dxs, dys = L.dys, dxy = dxs * dys;
    const double idxs = 1.0 / dxs, idys = 1.0 / dys;
    double g[4][4];
    double DxV0[2][2], DyV0[2][2], DxyV0[2][2];
    Patch B; double v, sx, sy;

    nbhd<0, 0>(P, field, L, g);                                   // V0 = S_f(z) on the neighbourhood
    fd_x(g, L, DxV0); fd_y(g, L, DyV0); fd_xy(g, L, DxyV0);
    // patch 1: f
    corners(g, B.F); copy22(DxV0, B.FX, dxs); copy22(DyV0, B.FY, dys); copy22(DxyV0, B.FXY, dxy);
    patch_eval<false, false>(B, hx, hy, v, sx, sy);
    out[0] = v;

    // patch 2: df/dx  (F = Dx V0, FX = Dx Vx dx, FY = Dxy V0 dy, FXY = Dxy Vx dx dy)
    nbhd<1, 0>(P, field, L, g);                                   // Vx = S_fx(z)
    copy22(DxV0, B.F, 1.0); copy22(DxyV0, B.FY, dys);
    fd_x(g, L, B.FX); scale22(B.FX, dxs);
    fd_xy(g, L, B.FXY); scale22(B.FXY, dxy);
    patch_eval<ORDER2, ORDER2>(B, hx, hy, v, sx, sy);
    out[1] = v;
    if(ORDER2){ out[4] = sx * idxs; out[7] = sy * idys; }

    // patch 3: df/dy  (F = Dy V0, FX = Dxy V0 dx, FY = Dy Vy dy, FXY = Dxy Vy dx dy)
    nbhd<2, 0>(P, field, L, g);                                   // Vy = S_fy(z)
    copy22(DyV0, B.F, 1.0); copy22(DxyV0, B.FX, dxs);
    fd_y(g, L, B.FY); scale22(B.FY, dys);
    fd_xy(g, L, B.FXY); scale22(B.FXY, dxy);
    patch_eval<false, ORDER2>(B, hx, hy, v, sx, sy);
    out[2] = v;
    if(ORDER2) out[5] = sy * idys;

    // patch 4: df/dz  (F = V0z, FX = Vxz dx, FY = Vyz dy, FXY = Dxy V0z dx dy)
    nbhd<0, 1>(P, field, L, g);                                   // V0z = S_f'(z)
    corners(g, B.F);
    fd_xy(g, L, B.FXY); scale22(B.FXY, dxy);
    nbhd_corners<1, 1>(P, field, L, B.FX); scale22(B.FX, dxs);
    nbhd_corners<2, 1>(P, field, L, B.FY); scale22(B.FY, dys);
    patch_eval<ORDER2, ORDER2>(B, hx, hy, v, sx, sy);
    out[3] = v;
    if(ORDER2){
        out[8] = sx * idxs; out[9] = sy * idys;
        // patch 5: d2f/dz2 (F = V0zz, FX = Dx V0zz dx, FY = Dy V0zz * DX (Q11, :1568-1571), FXY = Dxy V0zz dx dy)
        nbhd<0, 2>(P, field, L, g);
        corners(g, B.F);
        fd_x(g, L, B.FX); scale22(B.FX, dxs);
        fd_y(g, L, B.FY); scale22(B.FY, dxs);
        fd_xy(g, L, B.FXY); scale22(B.FXY, dxy);
        patch_eval<false, false>(B, hx, hy, v, sx, sy);
        out[6] = v;
    }
}

// Eval_Spline_f (:806-863): scalar value, y rows scaled by dx_scalar (Q11)
DEVINL double grid_eval_f(const GeoacDevParams& P, int field, const GridLoc& L){
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    double g[4][4]; Patch B; double v, sx, sy;
    nbhd<0, 0>(P, field, L, g);
    corners(g, B.F);
    fd_x(g, L, B.FX); scale22(B.FX, L.dxs);
    fd_y(g, L, B.FY); scale22(B.FY, L.dxs);
    fd_xy(g, L, B.FXY); scale22(B.FXY, L.dxs * L.dys);
    patch_eval<false, false>(B, hx, hy, v, sx, sy);
    return v;
}
// Eval_Spline_df(.., index = 2, ..) (:920-939): df/dz patch, y rows scaled by dx_scalar (Q11)
DEVINL double grid_eval_dfdz(const GeoacDevParams& P, int field, const GridLoc& L){
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    double g[4][4]; Patch B; double v, sx, sy;
    nbhd<0, 1>(P, field, L, g);
    corners(g, B.F);
    fd_xy(g, L, B.FXY); scale22(B.FXY, L.dxs * L.dys);
    nbhd_corners<1, 1>(P, field, L, B.FX); scale22(B.FX, L.dxs);
    nbhd_corners<2, 1>(P, field, L, B.FY); scale22(B.FY, L.dxs);
    patch_eval<false, false>(B, hx, hy, v, sx, sy);
    return v;
}

// scalar medium at a point (c(), u(), v(), rho() of G2S_MultiDimSpline3D.cpp:1633-1743, inputs clamped)
struct Medium3 { double c, u, v, rho, dcz, duz, dvz; };
template <bool WANT_RHO, bool WANT_DZ>
DEVINL Medium3 medium3_at(const GeoacDevParams& P, double x, double y, double z){
    double xe = clampd(x, P.gx[0], P.gx[P.gnx - 1]), ye = clampd(y, P.gy[0], P.gy[P.gny - 1]), ze = clampd(z, P.x_min, P.x_max);
    GridLoc L; grid_locate(P, xe, ye, ze, -1, L);
    Medium3 m;
    m.c = sqrt(kGamR * grid_eval_f(P, 0, L));
    m.u = grid_eval_f(P, 1, L);
    m.v = grid_eval_f(P, 2, L);
    m.rho = WANT_RHO ? grid_eval_f(P, 3, L) : 0.0;
    if(WANT_DZ){
        m.dcz = kGamR / (2.0 * m.c) * grid_eval_dfdz(P, 0, L);
        m.duz = grid_eval_dfdz(P, 1, L);
        m.dvz = grid_eval_dfdz(P, 2, L);
    } else { m.dcz = m.duz = m.dvz = 0.0; }
    return m;
}

// fused GeoAc_UpdateSources + GeoAc_EvalSrcEq of the range-dependent Cartesian set (EquationSets.3DRngDep.cpp:218-393)
// y: x, y, z, nu_x, nu_y, nu_z | X_th(3), mu_th(3) | X_ph(3), mu_ph(3)
template <bool AMP>
DEVINL void rngdep_rhs(const GeoacDevParams& P, int& kz, const double* y, double* dy){
    const double xe = clampd(y[0], P.gx[0], P.gx[P.gnx - 1]), ye = clampd(y[1], P.gy[0], P.gy[P.gny - 1]), ze = clampd(y[2], P.x_min, P.x_max);
    GridLoc L; grid_locate(P, xe, ye, ze, kz, L);
    kz = L.kz;
    double T[10], U[10], V[10];
    grid_eval_all<AMP>(P, 0, L, T);
    grid_eval_all<AMP>(P, 1, L, U);
    grid_eval_all<AMP>(P, 2, L, V);
    const double n0 = y[3], n1 = y[4], n2 = y[5];
    const double qT = kGamR * T[0];
    const double ic = frsq(qT);
    const double c = qT * ic;
    const double hc = (0.5 * kGamR) * ic;
    const double dc[3] = { hc * T[1], hc * T[2], hc * T[3] };
    const double nn = __builtin_fma(n0, n0, __builtin_fma(n1, n1, n2 * n2));
    const double inm = frsq(nn);
    const double numag = nn * inm;
    const double cn = c * inm;
    const double cg0 = __builtin_fma(cn, n0, U[0]), cg1 = __builtin_fma(cn, n1, V[0]), cg2 = cn * n2;
    const double icg = frsq(__builtin_fma(cg0, cg0, __builtin_fma(cg1, cg1, cg2 * cg2)));
    const double u0 = cg0 * icg, u1 = cg1 * icg, u2 = cg2 * icg;
    double Hn[3];
    #pragma unroll
    for(int i = 0; i < 3; i++) Hn[i] = __builtin_fma(numag, dc[i], __builtin_fma(n0, U[1 + i], n1 * V[1 + i]));
    dy[0] = u0; dy[1] = u1; dy[2] = u2;
    #pragma unroll
    for(int i = 0; i < 3; i++) dy[3 + i] = -icg * Hn[i];
    if(AMP){
        // symmetric second-derivative matrices: index [n][m] from out[4..9] = xx, yy, zz, xy, xz, yz
        const int ij[3][3] = { {4, 7, 8}, {7, 5, 9}, {8, 9, 6} };
        const double hc3 = (0.25 * kGamR * kGamR) * (ic * ic * ic);                // gamR^2 / (4 c^3)
        #pragma unroll
        for(int a = 0; a < 2; a++){
            const double X[3] = { y[6 + 6 * a], y[7 + 6 * a], y[8 + 6 * a] };
            const double m[3] = { y[9 + 6 * a], y[10 + 6 * a], y[11 + 6 * a] };
            double dc3 = 0.0, du3 = 0.0, dv3 = 0.0;
            #pragma unroll
            for(int n = 0; n < 3; n++){ dc3 = __builtin_fma(X[n], dc[n], dc3); du3 = __builtin_fma(X[n], U[1 + n], du3); dv3 = __builtin_fma(X[n], V[1 + n], dv3); }
            const double dnu = __builtin_fma(n0, m[0], __builtin_fma(n1, m[1], n2 * m[2])) * inm;
            const double al = inm * __builtin_fma(-cn, dnu, dc3);
            const double dcg0 = __builtin_fma(n0, al, __builtin_fma(cn, m[0], du3));
            const double dcg1 = __builtin_fma(n1, al, __builtin_fma(cn, m[1], dv3));
            const double dcg2 = __builtin_fma(n2, al, cn * m[2]);
            const double e = icg * __builtin_fma(u0, dcg0, __builtin_fma(u1, dcg1, u2 * dcg2));
            dy[6 + 6 * a] = __builtin_fma(icg, dcg0, -u0 * e);
            dy[7 + 6 * a] = __builtin_fma(icg, dcg1, -u1 * e);
            dy[8 + 6 * a] = __builtin_fma(icg, dcg2, -u2 * e);
            #pragma unroll
            for(int i = 0; i < 3; i++){
                double ddc = 0.0, ddu = 0.0, ddv = 0.0;
                #pragma unroll
                for(int mm = 0; mm < 3; mm++){
                    ddc = __builtin_fma(X[mm], __builtin_fma(hc, T[ij[i][mm]], -hc3 * (T[1 + i] * T[1 + mm])), ddc);
                    ddu = __builtin_fma(X[mm], U[ij[i][mm]], ddu);
                    ddv = __builtin_fma(X[mm], V[ij[i][mm]], ddv);
                }
                dy[9 + 6 * a + i] = icg * (e * Hn[i]
                                           - (dnu * dc[i] + numag * ddc + m[0] * U[1 + i] + m[1] * V[1 + i] + n0 * ddu + n1 * ddv));
            }
        }
    }
}

#endif
