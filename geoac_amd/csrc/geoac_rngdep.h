// geoac_rngdep.h - device code of the range-dependent Cartesian set (GeoAc3D.RngDep): included by geoac_kernels.hip.
//
// Atmosphere = grid of vertical profiles.  The reference (Code/Atmo/G2S_MultiDimSpline3D.cpp) keeps, per node (i, j), natural
// cubic splines in z of f, of the centred-difference df/dx and of df/dy, and evaluates a field at (x, y, z) as a bicubic
// Hermite patch over the cell whose corner data (f, f_x, f_y, f_xy) are vertical-spline values and horizontal finite
// differences of vertical-spline values; every derivative the equation set needs is again its own bicubic patch of
// differentiated data (Eval_Spline_AllOrder1/2, :1156-1593).
//
// The finite differences the reference takes at evaluation time (BiCubic_Deriv_*, :568-800) are centred at a NODE (clamped
// to one-sided at the grid edge) and linear in the vertical-spline coefficients, and all nodes share the z grid.  So the
// host differences the COEFFICIENTS once (geoac_upload_atmo_3d) and the device table holds, per (field, kz, node), ten
// vertical cubics  F, DxF, DyF, DxyF, Vx, DxVx, DxyVx, Vy, DyVy, DxyVy  (V0 = S_f, Vx = S_fx, Vy = S_fy of the reference)
// in the derivative-friendly form (c0, c1, 2c2, 6c3): one evaluation touches the FOUR cell corners only, 320 contiguous
// bytes each, instead of a 4x4 neighbourhood of scattered records - the path is bound by divergent table gathers.
// The 16x16 matrix product + power sums of the reference are evaluated as the equivalent tensor Hermite form, accumulated
// corner by corner.  Quirk Q11 kept: the scalar evaluators and the d2f/dz2 patch scale the y-derivative rows by the x cell size.
//
// The spherical set (GeoAcGlobal.RngDep, Code/Atmo/G2S_GlobalMultiDimSpline3D.cpp) uses the same scheme with x = latitude,
// y = longitude [rad], z = geocentric radius and its own quirks (Q12; template flag GLB): the host table already carries the
// reference's S_fx / S_fy slope systems and the truncated d/dr of those splines; here its scalar evaluators and d2f/dr2 patch
// scale the y rows by the y cell size, and AllOrder2's mixed second derivatives stay in SCALED cell coordinates.
#ifndef GEOAC_RNGDEP_H_
#define GEOAC_RNGDEP_H_
#include <utility>
#include <type_traits>

#define GEOAC_GREC 40        // doubles per (field, kz, node) record of T, u, v: ten cubics
#define GEOAC_GREC_RHO 16    // rho: F, DxF, DyF, DxyF only (scalar evaluator)
enum { GC_F = 0, GC_DXF, GC_DYF, GC_DXYF, GC_VX, GC_DXVX, GC_DXYVX, GC_VY, GC_DYVY, GC_DXYVY };
// Cartesian set: Vx and DxF (Vy and DyF) are the same cubic up to rounding - the natural-spline systems are linear and every node has
// the same z grid, so the spline of the differences is the difference of the splines (< 1e-10 of the coefficient scale, k_gb_pack8) -
// and the kernels read a PACKED table of eight cubics  F, DxF, DyF, DxyF | DxVx, DxyVx, DyVy, DxyVy:  256 bytes, line-aligned, two
// cache lines per record instead of 320 bytes over three or four.  (The spherical set's S_fx / S_fy carry the Q12 terms: ten cubics.)
#ifndef GEOAC_GREC_CART
#define GEOAC_GREC_CART 32   // 40: the Cartesian kernels read the full records as well (A/B builds)
#endif
template <bool GLB> struct GRec {
    static constexpr bool PACKED = !GLB && GEOAC_GREC_CART == 32;
    static constexpr int  N = PACKED ? 32 : GEOAC_GREC, NCUB = N / 4;
    static constexpr int  VX = PACKED ? (int)GC_DXF : (int)GC_VX, VY = PACKED ? (int)GC_DYF : (int)GC_VY;
    static constexpr int  DXVX = PACKED ? 4 : (int)GC_DXVX, DXYVX = PACKED ? 5 : (int)GC_DXYVX, DYVY = PACKED ? 6 : (int)GC_DYVY, DXYVY = PACKED ? 7 : (int)GC_DXYVY;
};

struct GridLoc {
    int kz;                 // vertical segment
    int ny;                 // nodes per x row
    int n00, n01, n10, n11; // node index (ix*ny + iy) of the cell corners n<a><b>, a <-> x edge, b <-> y edge (scalars, not an array: a rolled
                            // corner loop would index an array dynamically and put it in scratch)
    double t;               // z - z0[kz]
    double xs, ys;          // position inside the cell, scaled to [0, 1]
    double dxs, dys;        // cell sizes (dx_scalar, dy_scalar)
    double idxs, idys;      // and their reciprocals (frcp: < 1 ulp)
};

// locate (x, y, z) (already clamped to the grid): cell, corners, vertical segment.  kz_hint < 0: search from scratch.
// kxy (the RK4 kernels): the cell of this ray's previous evaluation, kx << 16 | ky, or -1.  A ray stays in one cell for thousands of stages:
// with the hint the node coordinates of the cell are four loads issued together (one trip to memory, beside the z nodes') and a range test;
// the scan over the nodes - one DEPENDENT trip per node, hipcc cannot take them through the scalar cache - runs only when the ray has
// left the cell.  (Measured on the four-lane kernel of the eigenray rounds: the scans were nine round trips of a stage's ~8000 cycles.)
DEVINL void grid_locate(const GeoacDevParams& P, double x, double y, double z, int kz_hint, GridLoc& L, const double* __restrict__ gzp = nullptr, int* kxy = nullptr, double* cell = nullptr){
    const double* __restrict__ gz = gzp ? gzp : P.gz;              // the z nodes: global table, or the kernel's LDS copy (record-cache kernels)
    const int nx = P.gnx, ny = P.gny;
    int kx = 0, ky = 0;
    double X1 = 0.0, X2 = 0.0, Y1 = 0.0, Y2 = 0.0;
    bool found = false;
    if(kxy && *kxy >= 0){
        kx = *kxy >> 16; ky = *kxy & 0xffff;
        if(cell){ X1 = cell[0]; X2 = cell[1]; Y1 = cell[2]; Y2 = cell[3]; }       // (the hinted cell's node coordinates, kept by the caller)
        else { X1 = P.gx[kx]; X2 = P.gx[kx + 1]; Y1 = P.gy[ky]; Y2 = P.gy[ky + 1]; }
        // the scan's answer is kx iff x lies in [X1, X2) - the first cell also takes what is below it, the last one what is above
        found = (x >= X1 || kx == 0) && (x < X2 || kx == nx - 2) && (y >= Y1 || ky == 0) && (y < Y2 || ky == ny - 2);
    }
    if(!found){
        kx = 0; ky = 0;
        for(int i = 1; i < nx - 1; i++) kx += (x >= P.gx[i]) ? 1 : 0;   // last i with x >= X[i], capped at nx-2 (branch-free scan)
        for(int j = 1; j < ny - 1; j++) ky += (y >= P.gy[j]) ? 1 : 0;
        X1 = P.gx[kx]; X2 = P.gx[kx + 1]; Y1 = P.gy[ky]; Y2 = P.gy[ky + 1];
        if(kxy) *kxy = (nx < 32768 && ny < 32768) ? ((kx << 16) | ky) : -1;
        if(cell){ cell[0] = X1; cell[1] = X2; cell[2] = Y1; cell[3] = Y2; }
    }
    int kz;
    if(kz_hint < 0){
        double span = P.x_max - P.x_min;                                  // x_min/x_max hold the z range for this set
        kz = (int)((z - P.x_min) / span * (double)P.nseg);
        kz = kz < 0 ? 0 : (kz > P.nseg - 1 ? P.nseg - 1 : kz);
    } else kz = kz_hint;
    while(kz > 0 && z < gz[kz]) kz--;
    while(kz < P.nseg - 1 && z > gz[kz + 1]) kz++;
    L.kz = kz;
    L.t = z - gz[kz];
    L.ny = ny;
    L.n00 = kx * ny + ky;       L.n01 = kx * ny + ky + 1;
    L.n10 = (kx + 1) * ny + ky; L.n11 = (kx + 1) * ny + ky + 1;
    L.dxs = X2 - X1; L.dys = Y2 - Y1;
    L.idxs = frcp(L.dxs); L.idys = frcp(L.dys);
    L.xs = (x - X1) * L.idxs; L.ys = (y - Y1) * L.idys;
}

// Hermite basis on [0,1] and its derivative: value weights h[0], h[1] and slope weights g[0], g[1] of the two cell edges
struct Herm { double h[2], g[2], dh[2], dg[2]; };
DEVINL Herm hermite(double s){
    // explicit FMAs, contraction off: the evaluators below are inlined into differently shaped code (per-lane, cooperative, LDS-DMA) and
    // must give a ray the same bits whichever of them serves it - hipcc's own choice of what to fuse varies with the surroundings
    #pragma clang fp contract(off)
    Herm H; const double s2 = s * s, s3 = s2 * s;
    H.h[1] = __builtin_fma(-2.0, s3, 3.0 * s2); H.h[0] = 1.0 - H.h[1];
    H.g[0] = __builtin_fma(-2.0, s2, s) + s3;   H.g[1] = s3 - s2;
    H.dh[1] = 6.0 * (s - s2);                   H.dh[0] = -H.dh[1];
    H.dg[0] = __builtin_fma(3.0, s2, __builtin_fma(-4.0, s, 1.0)); H.dg[1] = __builtin_fma(3.0, s2, -2.0 * s);
    return H;
}

// one vertical cubic (c0, c1, 2 c2, 6 c3), 32-byte aligned: two 16-byte loads
struct Cub { double c0, c1, d2, e3; };
typedef __attribute__((address_space(3))) char geoac_lds_char;
typedef double geoac_d2 __attribute__((ext_vector_type(2)));       // one 16-byte chunk (native vector: stays in registers)
DEVINL Cub load_cubic(const double* c){
    const geoac_d2* q = (const geoac_d2*)__builtin_assume_aligned(c, 16);
    const geoac_d2 lo = q[0], hi = q[1];
    return Cub{ lo.x, lo.y, hi.x, hi.y };
}
typedef __attribute__((address_space(3))) const geoac_d2 geoac_lds_d2;
DEVINL Cub load_cubic_lds(unsigned lds_off){                           // the same from an LDS byte offset (explicit ds_read_b128)
    const geoac_lds_d2* q = (const geoac_lds_d2*)(size_t)lds_off;
    const geoac_d2 lo = q[0], hi = q[1];
    return Cub{ lo.x, lo.y, hi.x, hi.y };
}
// f = c0 + t (c1 + t/2 (d2 + t/3 e3)), f' = c1 + t (d2 + t/2 e3): three and two FMAs (t/2, t/3 are common to all cubics of an evaluation)
DEVINL double cub_val(const Cub& c, double t, double t6){ const double th = 3.0 * t6, t3 = 2.0 * t6; return __builtin_fma(t, __builtin_fma(th, __builtin_fma(t3, c.e3, c.d2), c.c1), c.c0); }
DEVINL double cub_d1(const Cub& c, double t, double th){ return __builtin_fma(t, __builtin_fma(th, c.e3, c.d2), c.c1); }
DEVINL double cub_d2(const Cub& c, double t){ return __builtin_fma(t, c.e3, c.d2); }

template <bool GLB>
DEVINL const double* grid_rec(const GeoacDevParams& P, int field, int kz, int node){
    const size_t nn = (size_t)(P.gnx * P.gny);
    if(field < 3) return P.gtab + (((size_t)field * P.nseg + kz) * nn + node) * GRec<GLB>::N;
    return P.gtab + (size_t)3 * P.nseg * nn * GRec<GLB>::N + ((size_t)kz * nn + node) * GEOAC_GREC_RHO;
}

// tensor Hermite weights of corner (a, b): value (W), d/dxs (D), d/dys (E); index hh, gh, hg, gg <-> F, FX, FY, FXY
struct CornerW { double W[4], D[4], E[4]; };
template <bool WANT_D, bool WANT_E>
DEVINL CornerW corner_weights(const Herm& hx, const Herm& hy, int a, int b){
    // (a, b) are loop variables of a ROLLED corner loop (one corner's loads in flight at a time keeps the live set in
    // registers): pick the edge with selects so that nothing is indexed dynamically
    const double xh = a ? hx.h[1] : hx.h[0], xg = a ? hx.g[1] : hx.g[0], xdh = a ? hx.dh[1] : hx.dh[0], xdg = a ? hx.dg[1] : hx.dg[0];
    const double yh = b ? hy.h[1] : hy.h[0], yg = b ? hy.g[1] : hy.g[0], ydh = b ? hy.dh[1] : hy.dh[0], ydg = b ? hy.dg[1] : hy.dg[0];
    CornerW w;
    w.W[0] = xh * yh; w.W[1] = xg * yh; w.W[2] = xh * yg; w.W[3] = xg * yg;
    if(WANT_D){ w.D[0] = xdh * yh; w.D[1] = xdg * yh; w.D[2] = xdh * yg; w.D[3] = xdg * yg; }
    if(WANT_E){ w.E[0] = xh * ydh; w.E[1] = xg * ydh; w.E[2] = xh * ydg; w.E[3] = xg * ydg; }
    return w;
}
// (arithmetic, not a select over n00..n11: with a lane-dependent corner hipcc turns the select into a 16-byte table in scratch, a trip to
// memory on the critical path of every stage of the four-lane kernels)
DEVINL int corner_node(const GridLoc& L, int a, int b){ return L.n00 + a * L.ny + b; }
// the select form, for the LANE-dependent corner of the two- and four-lane kernels (there the arithmetic form costs the record-cache kernel
// 190 spilled registers and 5 % of its speed; the table it becomes is read once per stage)
DEVINL int corner_node_sel(const GridLoc& L, int a, int b){ return a ? (b ? L.n11 : L.n10) : (b ? L.n01 : L.n00); }
DEVINL double dot4(const double* w, double F, double FX, double FY, double FXY, double acc){
    return __builtin_fma(w[0], F, __builtin_fma(w[1], FX, __builtin_fma(w[2], FY, __builtin_fma(w[3], FXY, acc))));
}

// Eval_Spline_AllOrder1 (ORDER2 = false, :1156-1339) / AllOrder2 (:1341-1593):
// out = f, f_x, f_y, f_z [, f_xx, f_yy, f_zz, f_xy, f_xz, f_yz]
//   patch f    : F = V0,     FX = Dx V0 dx,    FY = Dy V0 dy,        FXY = Dxy V0 dx dy
//   patch f_x  : F = Dx V0,  FX = Dx Vx dx,    FY = Dxy V0 dy,       FXY = Dxy Vx dx dy     (its d/dx, d/dy: f_xx, f_xy)
//   patch f_y  : F = Dy V0,  FX = Dxy V0 dx,   FY = Dy Vy dy,        FXY = Dxy Vy dx dy     (its d/dy: f_yy)
//   patch f_z  : F = V0',    FX = Vx' dx,      FY = Vy' dy,          FXY = Dxy V0' dx dy    (its d/dx, d/dy: f_xz, f_yz)
//   patch f_zz : F = V0'',   FX = Dx V0'' dx,  FY = Dy V0'' dx (Q11, :1568-1571), FXY = Dxy V0'' dx dy
// butterfly sum over the four lanes of a quad (DPP quad_perm [1,0,3,2] then [2,3,0,1]); every lane ends with the bit-identical total
DEVINL double quad_sum(double v){
    int lo = __double2loint(v), hi = __double2hiint(v);
    double w = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true));
    v += w;
    lo = __double2loint(v); hi = __double2hiint(v);
    w = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true));
    return v + w;
}

DEVINL double pair_sum(double v){
    int lo = __double2loint(v), hi = __double2hiint(v);
    double w = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true));
    return v + w;
}

// NL = 2 or 4 lanes per ray (small fans).  Lane cq of the group evaluates 4/NL of the cell corners and the partial sums are added
// across the group: 3 (NL = 4) or 6 (NL = 2) dependent gather batches per RHS instead of 12 (the kernel is bound by their latency).
// rec_lds (four-lane kernels with a record cache): this lane's record of `field` for its corner, already in LDS
// LDSOFF (eight-lane kernel): the record is read from LDS byte offset lds_off
template <bool ORDER2, bool GLB, int NL = 1, bool LDSOFF = false>
DEVINL void grid_eval_all(const GeoacDevParams& P, int field, const GridLoc& L, double* out, int cq = 0, const double* rec_lds = nullptr, unsigned lds_off = 0){
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    const double dxs = L.dxs, dys = L.dys, dxy = dxs * dys;
    const double t = L.t, th = 0.5 * t, t6 = t * (1.0 / 6.0);
    typedef GRec<GLB> R;
    const double* __restrict__ base = grid_rec<GLB>(P, field, L.kz, 0);
    double o[10];
    #pragma unroll
    for(int i = 0; i < 10; i++) o[i] = 0.0;
    #pragma unroll 2                                               // two corners per gather batch (one: 12 dependent batches per RHS; four: more spills than it saves)
    for(int cn = cq * (4 / NL); cn < (cq + 1) * (4 / NL); cn++){
        {
            const int a = cn >> 1, b = cn & 1;
            const double* r = rec_lds ? rec_lds : base + (size_t)corner_node(L, a, b) * R::N;      // (by arithmetic: a select over n00 .. n11 with a lane-dependent corner becomes a table in scratch memory)
            Cub c[R::NCUB];                                        // all loads of the corner in flight before the first use
            #pragma unroll
            for(int i = 0; i < R::NCUB; i++) c[i] = LDSOFF ? load_cubic_lds(lds_off + 32u * i) : load_cubic(r + 4 * i);
            CornerW w = corner_weights<ORDER2, ORDER2>(hx, hy, a, b);
            // fold the cell-size factors of the FX / FY / FXY rows into the weights
            const double Wq = w.W[2] * (GLB ? dys : dxs);          // Cartesian: Q11 row of the f_zz patch (y row scaled by dx)
            w.W[1] *= dxs; w.W[2] *= dys; w.W[3] *= dxy;
            if(ORDER2){ w.D[1] *= dxs; w.D[2] *= dys; w.D[3] *= dxy; w.E[1] *= dxs; w.E[2] *= dys; w.E[3] *= dxy; }
            const double F = cub_val(c[GC_F], t, t6), DxF = cub_val(c[GC_DXF], t, t6), DyF = cub_val(c[GC_DYF], t, t6), DxyF = cub_val(c[GC_DXYF], t, t6);
            const double DxVx = cub_val(c[R::DXVX], t, t6), DxyVx = cub_val(c[R::DXYVX], t, t6);
            const double DyVy = cub_val(c[R::DYVY], t, t6), DxyVy = cub_val(c[R::DXYVY], t, t6);
            const double Fz = cub_d1(c[GC_F], t, th), Vxz = cub_d1(c[R::VX], t, th), Vyz = cub_d1(c[R::VY], t, th), DxyFz = cub_d1(c[GC_DXYF], t, th);
            o[0] = dot4(w.W, F, DxF, DyF, DxyF, o[0]);
            o[1] = dot4(w.W, DxF, DxVx, DxyF, DxyVx, o[1]);
            o[2] = dot4(w.W, DyF, DxyF, DyVy, DxyVy, o[2]);
            o[3] = dot4(w.W, Fz, Vxz, Vyz, DxyFz, o[3]);
            if(ORDER2){
                o[4] = dot4(w.D, DxF, DxVx, DxyF, DxyVx, o[4]);    // d/dxs of the f_x patch
                o[7] = dot4(w.E, DxF, DxVx, DxyF, DxyVx, o[7]);    // d/dys of the f_x patch
                o[5] = dot4(w.E, DyF, DxyF, DyVy, DxyVy, o[5]);    // d/dys of the f_y patch
                o[8] = dot4(w.D, Fz, Vxz, Vyz, DxyFz, o[8]);       // d/dxs of the f_z patch
                o[9] = dot4(w.E, Fz, Vxz, Vyz, DxyFz, o[9]);       // d/dys of the f_z patch
                const double Fzz = cub_d2(c[GC_F], t), DxFzz = cub_d2(c[GC_DXF], t), DyFzz = cub_d2(c[GC_DYF], t), DxyFzz = cub_d2(c[GC_DXYF], t);
                o[6] = __builtin_fma(w.W[0], Fzz, __builtin_fma(w.W[1], DxFzz, __builtin_fma(Wq, DyFzz, __builtin_fma(w.W[3], DxyFzz, o[6]))));
            }
        }
    }
    if(NL > 1){
        #pragma unroll
        for(int i = 0; i < (ORDER2 ? 10 : 4); i++) o[i] = (NL == 4) ? quad_sum(o[i]) : pair_sum(o[i]);
    }
    if(ORDER2 && !GLB){                                            // spherical set: left in scaled coordinates (Q12c, :1328-1338, :1374-1384, :1420-1424)
        const double idxs = L.idxs, idys = L.idys;
        o[4] *= idxs; o[8] *= idxs; o[7] *= idys; o[5] *= idys; o[9] *= idys;
    }
    #pragma unroll
    for(int i = 0; i < (ORDER2 ? 10 : 4); i++) out[i] = o[i];
}

// ---- wave-cooperative record gather (one ray per lane, large fans) --------------------------------------------------------------
// With one record per lane every global_load_dwordx4 of a wave touches 64 different cache lines and the CU's texture-address / L1 path
// is busy 48 cycles per wave-instruction, against 17-23 when lanes read contiguous bytes (tools/ubench_gather.hip; 240 such loads per
// RK4 stage).  Here the four lanes of a QUAD fetch 64 contiguous bytes per instruction (16 pieces per instruction instead of 64 lines):
// a quad walks through the half-records (160 B: five cubics) of its four lanes back to back, 10 loads for 4 x 160 B, and the chunks are
// handed to their owners through LDS: every lane stores what it loaded at slot[owner][chunk] and then reads its own 10 chunks.  Slots are
// 176 B apart (160 B + 16 B pad): the ds_read_b128 of a wave are bank-conflict free (bank group (11 slot + chunk) mod 16 is distinct over
// each 16-lane access group), the ds_write_b128 too except the two of ten that straddle two owners (2-way).
// All 64 lanes of the wave take part, also lanes whose ray has finished (they fetch for their quad mates): the caller keeps the wave
// converged around this function.  LDS: 64 x 176 B = 11 KiB per wave.  The next half-record's loads are in flight while one is evaluated.
// Same operations in the same order as grid_eval_all: a ray's numbers do not depend on which of the two gathers served it.
// Measured (config-4 share, MI355X): texture-path busy cycles -32 %, L1 accesses -38 %; 5x5x1400 grid (38 MB table) 2.94 -> 2.50 s,
// 5x5x350 grid (9.6 MB) 2.16 -> 2.14 s: with one wave per SIMD the 480 LDS instructions per stage (ds_write_b128: 13 issue cycles)
// cost the wave what the gathers cost the texture path.  Tried and dropped: fetching only the (at most two) DISTINCT records of a quad -
// neighbouring rays fall out of step after their first ground reflection, 20 % of the lanes then need a third record.
#define GEOAC_COOP_SLOT 176
#ifndef GEOAC_COOP_GLDS
#define GEOAC_COOP_GLDS 1             // Cartesian cooperative kernels: gather by LDS-DMA (grid_eval3_glds); 0: register-staged (grid_eval3_coop8)
#endif
#ifndef GEOAC_COOP_WAVES
#define GEOAC_COOP_WAVES 1            // waves per SIMD the cooperative kernels are compiled for
#endif
template <int O> DEVINL unsigned quad_bcast_u32(unsigned v){ return (unsigned)__builtin_amdgcn_mov_dpp((int)v, O * 0x55, 0xF, 0xF, true); }

template <bool ORDER2, bool GLB>
DEVINL void grid_eval3_coop(const GeoacDevParams& P, const GridLoc& L, double (*M)[10], char* ldsw){
    const unsigned lane = threadIdx.x & 63u, r = lane & 3u;
    const bool hi2 = r >= 2u;                                                         // the upper lane pair of the quad (loads that straddle two owners)
    const unsigned nn = (unsigned)(P.gnx * P.gny);
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    const double dxs = L.dxs, dys = L.dys, dxy = dxs * dys;
    const double t = L.t, th = 0.5 * t, t6 = t * (1.0 / 6.0);
    const size_t fstride = (size_t)P.nseg * nn * (GEOAC_GREC * sizeof(double));       // bytes per field block of the table
    // the quad's 40 chunks of a half-record round are numbered g = 10 owner + chunk; load j of lane r fetches g = 4 j + r:
    //   global address = field block + [off_owner - 160 owner + 16 r] + 64 j + 160 half     LDS address = [176 quad + 16 r] + 16 owner + 64 j
    //   (16 chunk = 64 j + 16 r - 160 owner;  slot = 176 (quad + owner) + 16 chunk).  The - 160 owner term is kept non-negative by a bias
    //   of 480 that the base pointer takes back: the offsets stay unsigned 32-bit.
    const char* __restrict__ tabb = (const char*)P.gtab - 480;
    const unsigned rb = 16u * r + 480u;
    char* const wq = ldsw + (lane & ~3u) * GEOAC_COOP_SLOT + 16u * r;
    const char* const rslot = ldsw + lane * GEOAC_COOP_SLOT;
    #pragma unroll
    for(int f = 0; f < 3; f++){
        #pragma unroll
        for(int i = 0; i < 10; i++) M[f][i] = 0.0;
    }
    // owner of load j: j = 0, 1 -> 0; 2 -> 0 | 1; 3, 4 -> 1; 5, 6 -> 2; 7 -> 2 | 3; 8, 9 -> 3
    #define GEOAC_COOP_OWNER(j, B) ((j) < 2 ? B[0] : (j) == 2 ? (hi2 ? B[1] : B[0]) : (j) < 5 ? B[1] : (j) < 7 ? B[2] : (j) == 7 ? (hi2 ? B[3] : B[2]) : B[3])
    unsigned gb[4];                                                                   // global bases of the quad's owners for the corner being fetched
    const unsigned wb[4] = { 0u, 16u, 32u, 48u };
    {
        const unsigned off = ((unsigned)L.kz * nn + (unsigned)corner_node(L, 0, 0)) * (unsigned)(GEOAC_GREC * sizeof(double));
        gb[0] = quad_bcast_u32<0>(off) + rb; gb[1] = quad_bcast_u32<1>(off) + (rb - 160u); gb[2] = quad_bcast_u32<2>(off) + (rb - 320u); gb[3] = quad_bcast_u32<3>(off) + (rb - 480u);
    }
    geoac_d2 v[10];                                                                   // in flight: the NEXT half-record's chunks
    #pragma unroll
    for(int j = 0; j < 10; j++) v[j] = *(const geoac_d2*)(tabb + GEOAC_COOP_OWNER(j, gb) + 64 * j);
    double cF1 = 0, cF2 = 0, cF3 = 0, cFz = 0, cVxz = 0, cFxyz = 0;                   // carried from the first half of a record to the second
    #pragma unroll
    for(int h = 0; h < 24; h++){
        const int n = h >> 1, part = h & 1, cn = n / 3, f = n % 3, a = cn >> 1, b = cn & 1;
        #pragma unroll
        for(int j = 0; j < 10; j++) *(geoac_d2*)(wq + GEOAC_COOP_OWNER(j, wb) + 64 * j) = v[j];
        if(h + 1 < 24){
            const int n1 = (h + 1) >> 1, part1 = (h + 1) & 1, cn1 = n1 / 3, f1 = n1 % 3;
            if(part1 == 0 && f1 == 0){                                                // first half-record of the next corner: its owners' bases
                const unsigned off = ((unsigned)L.kz * nn + (unsigned)corner_node(L, cn1 >> 1, cn1 & 1)) * (unsigned)(GEOAC_GREC * sizeof(double));
                gb[0] = quad_bcast_u32<0>(off) + rb; gb[1] = quad_bcast_u32<1>(off) + (rb - 160u); gb[2] = quad_bcast_u32<2>(off) + (rb - 320u); gb[3] = quad_bcast_u32<3>(off) + (rb - 480u);
            }
            const char* __restrict__ fb = tabb + (size_t)f1 * fstride + 160 * part1;
            #pragma unroll
            for(int j = 0; j < 10; j++) v[j] = *(const geoac_d2*)(fb + GEOAC_COOP_OWNER(j, gb) + 64 * j);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        Cub c[5];
        #pragma unroll
        for(int i = 0; i < 5; i++){
            const geoac_d2 lo = *(const geoac_d2*)(rslot + 32 * i), hi = *(const geoac_d2*)(rslot + 32 * i + 16);
            c[i] = Cub{ lo.x, lo.y, hi.x, hi.y };
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");              // the next half-record's stores stay behind these reads
        __builtin_amdgcn_wave_barrier();
        CornerW w = corner_weights<ORDER2, ORDER2>(hx, hy, a, b);
        const double Wq = w.W[2] * (GLB ? dys : dxs);                                 // Cartesian: Q11 row of the f_zz patch (y row scaled by dx)
        w.W[1] *= dxs; w.W[2] *= dys; w.W[3] *= dxy;
        if(ORDER2){ w.D[1] *= dxs; w.D[2] *= dys; w.D[3] *= dxy; w.E[1] *= dxs; w.E[2] *= dys; w.E[3] *= dxy; }
        double* o = M[f];
        if(part == 0){                                                                // F, DxF, DyF, DxyF, Vx
            const double F = cub_val(c[GC_F], t, t6), DxF = cub_val(c[GC_DXF], t, t6), DyF = cub_val(c[GC_DYF], t, t6), DxyF = cub_val(c[GC_DXYF], t, t6);
            const double Fz = cub_d1(c[GC_F], t, th), Vxz = cub_d1(c[GC_VX], t, th), DxyFz = cub_d1(c[GC_DXYF], t, th);
            o[0] = dot4(w.W, F, DxF, DyF, DxyF, o[0]);
            if(ORDER2){
                const double Fzz = cub_d2(c[GC_F], t), DxFzz = cub_d2(c[GC_DXF], t), DyFzz = cub_d2(c[GC_DYF], t), DxyFzz = cub_d2(c[GC_DXYF], t);
                o[6] = __builtin_fma(w.W[0], Fzz, __builtin_fma(w.W[1], DxFzz, __builtin_fma(Wq, DyFzz, __builtin_fma(w.W[3], DxyFzz, o[6]))));
            }
            cF1 = DxF; cF2 = DyF; cF3 = DxyF; cFz = Fz; cVxz = Vxz; cFxyz = DxyFz;
        } else {                                                                      // DxVx, DxyVx, Vy, DyVy, DxyVy
            const double DxVx = cub_val(c[GC_DXVX - 5], t, t6), DxyVx = cub_val(c[GC_DXYVX - 5], t, t6);
            const double DyVy = cub_val(c[GC_DYVY - 5], t, t6), DxyVy = cub_val(c[GC_DXYVY - 5], t, t6);
            const double Vyz = cub_d1(c[GC_VY - 5], t, th);
            const double DxF = cF1, DyF = cF2, DxyF = cF3, Fz = cFz, Vxz = cVxz, DxyFz = cFxyz;
            o[1] = dot4(w.W, DxF, DxVx, DxyF, DxyVx, o[1]);
            o[2] = dot4(w.W, DyF, DxyF, DyVy, DxyVy, o[2]);
            o[3] = dot4(w.W, Fz, Vxz, Vyz, DxyFz, o[3]);
            if(ORDER2){
                o[4] = dot4(w.D, DxF, DxVx, DxyF, DxyVx, o[4]);
                o[7] = dot4(w.E, DxF, DxVx, DxyF, DxyVx, o[7]);
                o[5] = dot4(w.E, DyF, DxyF, DyVy, DxyVy, o[5]);
                o[8] = dot4(w.D, Fz, Vxz, Vyz, DxyFz, o[8]);
                o[9] = dot4(w.E, Fz, Vxz, Vyz, DxyFz, o[9]);
            }
        }
    }
    #undef GEOAC_COOP_OWNER
    if(ORDER2 && !GLB){                                                               // spherical set: left in scaled coordinates (Q12c)
        const double idxs = L.idxs, idys = L.idys;
        #pragma unroll
        for(int f = 0; f < 3; f++){ M[f][4] *= idxs; M[f][8] *= idxs; M[f][7] *= idys; M[f][5] *= idys; M[f][9] *= idys; }
    }
}

// The same gather over the PACKED Cartesian records (256 B, line-aligned): half-records of four cubics = one 128-byte cache line.  A quad
// walks through the four half-records of its lanes in 8 loads (load j: owner j >> 1, chunks 4 (j & 1) + r; every quad-load is 64 bytes
// inside ONE line), LDS slots 144 B apart (128 B + 16 B pad: reads and writes of a wave bank-conflict free), 8 loads / 8 ds_write /
// 8 ds_read per half-record instead of 10 / 10 / 10.  First half: F, DxF, DyF, DxyF -> f, f_z, f_zz, f_xz, f_yz; second half:
// DxVx, DxyVx, DyVy, DxyVy -> f_x, f_y, f_xx, f_xy, f_yy.  Offsets are unsigned 32-bit: the launch takes this kernel for tables < 4 GiB.
#define GEOAC_COOP8_SLOT 144
template <bool ORDER2>
DEVINL void grid_eval3_coop8(const GeoacDevParams& P, const GridLoc& L, double (*M)[10], char* ldsw){
    const unsigned lane = threadIdx.x & 63u, r = lane & 3u;
    const unsigned nn = (unsigned)(P.gnx * P.gny);
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    const double dxs = L.dxs, dys = L.dys, dxy = dxs * dys;
    const double t = L.t, th = 0.5 * t, t6 = t * (1.0 / 6.0);
    const size_t fstride = (size_t)P.nseg * nn * 256u;                                // bytes per field block of the table
    const char* __restrict__ tabb = (const char*)P.gtab;
    char* const wq = ldsw + (lane & ~3u) * GEOAC_COOP8_SLOT + 16u * r;
    const char* const rslot = ldsw + lane * GEOAC_COOP8_SLOT;
    #pragma unroll
    for(int f = 0; f < 3; f++){
        #pragma unroll
        for(int i = 0; i < 10; i++) M[f][i] = 0.0;
    }
    unsigned gb[4];                                                                   // byte offsets of the quad's owners' records (+ this lane's 16 r)
    {
        const unsigned off = ((unsigned)L.kz * nn + (unsigned)corner_node(L, 0, 0)) * 256u;
        gb[0] = quad_bcast_u32<0>(off) + 16u * r; gb[1] = quad_bcast_u32<1>(off) + 16u * r; gb[2] = quad_bcast_u32<2>(off) + 16u * r; gb[3] = quad_bcast_u32<3>(off) + 16u * r;
    }
    geoac_d2 v[8];                                                                    // in flight: the NEXT half-record's chunks
    #pragma unroll
    for(int j = 0; j < 8; j++) v[j] = *(const geoac_d2*)(tabb + gb[j >> 1] + 64 * (j & 1));
    double cDxF = 0, cDyF = 0, cDxyF = 0;                                             // carried from the first half of a record to the second
    #pragma unroll
    for(int h = 0; h < 24; h++){
        const int n = h >> 1, part = h & 1, cn = n / 3, f = n % 3, a = cn >> 1, b = cn & 1;
        #pragma unroll
        for(int j = 0; j < 8; j++) *(geoac_d2*)(wq + GEOAC_COOP8_SLOT * (j >> 1) + 64 * (j & 1)) = v[j];
        if(h + 1 < 24){
            const int n1 = (h + 1) >> 1, part1 = (h + 1) & 1, cn1 = n1 / 3, f1 = n1 % 3;
            if(part1 == 0 && f1 == 0){                                                // first half-record of the next corner: its owners' offsets
                const unsigned off = ((unsigned)L.kz * nn + (unsigned)corner_node(L, cn1 >> 1, cn1 & 1)) * 256u;
                gb[0] = quad_bcast_u32<0>(off) + 16u * r; gb[1] = quad_bcast_u32<1>(off) + 16u * r; gb[2] = quad_bcast_u32<2>(off) + 16u * r; gb[3] = quad_bcast_u32<3>(off) + 16u * r;
            }
            const char* __restrict__ fb = tabb + (size_t)f1 * fstride + 128 * part1;
            #pragma unroll
            for(int j = 0; j < 8; j++) v[j] = *(const geoac_d2*)(fb + gb[j >> 1] + 64 * (j & 1));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        Cub c[4];
        #pragma unroll
        for(int i = 0; i < 4; i++){
            const geoac_d2 lo = *(const geoac_d2*)(rslot + 32 * i), hi = *(const geoac_d2*)(rslot + 32 * i + 16);
            c[i] = Cub{ lo.x, lo.y, hi.x, hi.y };
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");              // the next half-record's stores stay behind these reads
        __builtin_amdgcn_wave_barrier();
        CornerW w = corner_weights<ORDER2, ORDER2>(hx, hy, a, b);
        const double Wq = w.W[2] * dxs;                                               // Q11 row of the f_zz patch (y row scaled by dx)
        w.W[1] *= dxs; w.W[2] *= dys; w.W[3] *= dxy;
        if(ORDER2){ w.D[1] *= dxs; w.D[2] *= dys; w.D[3] *= dxy; w.E[1] *= dxs; w.E[2] *= dys; w.E[3] *= dxy; }
        double* o = M[f];
        if(part == 0){                                                                // F, DxF, DyF, DxyF
            const double F = cub_val(c[GC_F], t, t6), DxF = cub_val(c[GC_DXF], t, t6), DyF = cub_val(c[GC_DYF], t, t6), DxyF = cub_val(c[GC_DXYF], t, t6);
            const double Fz = cub_d1(c[GC_F], t, th), Vxz = cub_d1(c[GC_DXF], t, th), Vyz = cub_d1(c[GC_DYF], t, th), DxyFz = cub_d1(c[GC_DXYF], t, th);
            o[0] = dot4(w.W, F, DxF, DyF, DxyF, o[0]);
            o[3] = dot4(w.W, Fz, Vxz, Vyz, DxyFz, o[3]);
            if(ORDER2){
                o[8] = dot4(w.D, Fz, Vxz, Vyz, DxyFz, o[8]);
                o[9] = dot4(w.E, Fz, Vxz, Vyz, DxyFz, o[9]);
                const double Fzz = cub_d2(c[GC_F], t), DxFzz = cub_d2(c[GC_DXF], t), DyFzz = cub_d2(c[GC_DYF], t), DxyFzz = cub_d2(c[GC_DXYF], t);
                o[6] = __builtin_fma(w.W[0], Fzz, __builtin_fma(w.W[1], DxFzz, __builtin_fma(Wq, DyFzz, __builtin_fma(w.W[3], DxyFzz, o[6]))));
            }
            cDxF = DxF; cDyF = DyF; cDxyF = DxyF;
        } else {                                                                      // DxVx, DxyVx, DyVy, DxyVy
            const double DxVx = cub_val(c[0], t, t6), DxyVx = cub_val(c[1], t, t6), DyVy = cub_val(c[2], t, t6), DxyVy = cub_val(c[3], t, t6);
            const double DxF = cDxF, DyF = cDyF, DxyF = cDxyF;
            o[1] = dot4(w.W, DxF, DxVx, DxyF, DxyVx, o[1]);
            o[2] = dot4(w.W, DyF, DxyF, DyVy, DxyVy, o[2]);
            if(ORDER2){
                o[4] = dot4(w.D, DxF, DxVx, DxyF, DxyVx, o[4]);
                o[7] = dot4(w.E, DxF, DxVx, DxyF, DxyVx, o[7]);
                o[5] = dot4(w.E, DyF, DxyF, DyVy, DxyVy, o[5]);
            }
        }
    }
    if(ORDER2){
        const double idxs = L.idxs, idys = L.idys;
        #pragma unroll
        for(int f = 0; f < 3; f++){ M[f][4] *= idxs; M[f][8] *= idxs; M[f][7] *= idys; M[f][5] *= idys; M[f][9] *= idys; }
    }
}

// ---- the same gather by LDS-DMA (global_load_lds_dwordx4: table -> LDS, no staging registers, no ds_write) ---------------------------
// With one wave per SIMD nothing hides a gather's trip to L2 but the wave's own prefetch, and the register-staged forms above can keep
// only one round (32 VGPRs) in flight.  Here a ROUND is a quarter record - two cubics, 64 bytes - of each of the four lanes of every quad:
// four loads (load j serves owner j; the four lanes of a quad fetch the owner's 64 bytes, one line piece), each landing lane-linear in a
// 1-KiB slot of a ring of 5 rounds x 4 slots (20 KiB per wave).  Four rounds (16 loads) are in flight while one is read and evaluated.
// Lane r of a quad fetches chunk (r + j) & 3 of owner j, so that the owners' reads (ds_read_b128 of chunk c at 1024 o + 64 quad +
// 16 ((c - o) & 3)) are bank-conflict free.  The loads are asm (hipcc would drain every LDS-DMA before the next ds_read); their completion is
// counted here: s_waitcnt vmcnt(12) before round rho is read leaves the three younger rounds in flight.  A slot is refilled one whole
// iteration after it was read, behind a register dependence on what those reads returned.  Arithmetic: the statements of grid_eval_all, one dot4 per accumulator and corner
// in the same order - the same bits.
// Measured and dropped (round 3): a SHARED gather - the wave numbers its distinct (segment, cell) keys (26 of 64 on the tiled config-4 fan), a round
// fetches each KEY's quarter record once (lanes without a key masked off by EXEC, four loads per round so that the vmcnt arithmetic stays
// static) and every lane reads its key's 64 bytes.  Same bits, config-4 share 1.24 -> 1.71 s: a load instruction costs the texture path the
// same with 17 lanes as with 64, the numbering loop is ~29 instructions per key, and the reads of 64-byte key slots conflict in the banks.
// With the loads of absent keys skipped by scalar branches (vmcnt by a four-way branch per round) the allocator spilt 750 registers.
#define GEOAC_GLDS_RING 5
#define GEOAC_GLDS_BYTES (GEOAC_GLDS_RING * 4096)

DEVINL void glds_round(const char* base, unsigned o0, unsigned o1, unsigned o2, unsigned o3, unsigned lds_dst, double d0, double d1, double d2, double d3){
    // d0..d3: one word of each ds_read of the round that was read out of this slot - naming them as inputs makes hipcc wait for those reads
    // (a counted lgkmcnt: they are a whole iteration old) before the DMA may overwrite what they read
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %2, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "s"(base), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(lds_dst), "v"(d0), "v"(d1), "v"(d2), "v"(d3) : "memory", "scc");
}
template <class F, int... I> DEVINL void geoac_static_for(F&& f, std::integer_sequence<int, I...>){ (f(std::integral_constant<int, I>{}), ...); }
template <int N> DEVINL void glds_wait(){ asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <bool ORDER2, bool GLB = false>
DEVINL void grid_eval3_glds(const GeoacDevParams& P, const GridLoc& L, double (*M)[10], char* ldsw){
    // Cartesian set: packed records, 4 rounds of two cubics each (F, DxF | DyF, DxyF | DxVx, DxyVx | DyVy, DxyVy).  Spherical set: the full
    // 320-byte records (64-byte aligned: a quad's piece never straddles a line), 5 rounds (F, DxF | DyF, DxyF | Vx, DxVx | DxyVx, Vy | DyVy, DxyVy)
    constexpr int NR = GLB ? 5 : 4, NROUND = 12 * NR;
    constexpr unsigned RB = GLB ? 320u : 256u;
    const unsigned lane = threadIdx.x & 63u, r = lane & 3u, quad = lane >> 2;
    const unsigned nn = (unsigned)(P.gnx * P.gny);
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    const double dxs = L.dxs, dys = L.dys, dxy = dxs * dys;
    const double t = L.t, th = 0.5 * t, t6 = t * (1.0 / 6.0);
    const size_t fstride = (size_t)P.nseg * nn * RB;                                  // bytes per field block of the table
    const char* __restrict__ tabb = (const char*)P.gtab;
    const unsigned ring = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(geoac_lds_char*)ldsw);   // LDS byte address of the wave's ring, in an SGPR
    // this lane's chunk c of its own quarter record: slot r of the round, position (c - r) & 3 of its quad's 64 bytes
    const char* const rq = ldsw + 1024u * r + 64u * quad;
    const char* const ra0 = rq + 16u * ((0u - r) & 3u), * const ra1 = rq + 16u * ((1u - r) & 3u), * const ra2 = rq + 16u * ((2u - r) & 3u), * const ra3 = rq + 16u * ((3u - r) & 3u);
    #pragma unroll
    for(int f = 0; f < 3; f++){
        #pragma unroll
        for(int i = 0; i < 10; i++) M[f][i] = 0.0;
    }
    unsigned gb[4];                                                                   // byte offsets: owner j's record + the chunk this lane fetches for it
    #define GEOAC_GLDS_BASES(cnx) { \
        const unsigned off = ((unsigned)L.kz * nn + (unsigned)corner_node(L, (cnx) >> 1, (cnx) & 1)) * RB; \
        gb[0] = quad_bcast_u32<0>(off) + 16u * (r & 3u); gb[1] = quad_bcast_u32<1>(off) + 16u * ((r + 1u) & 3u); \
        gb[2] = quad_bcast_u32<2>(off) + 16u * ((r + 2u) & 3u); gb[3] = quad_bcast_u32<3>(off) + 16u * ((r + 3u) & 3u); }
    geoac_d2 p0 = { 0.0, 0.0 }, p1 = p0, p2 = p0, p3 = p0;                            // what the previous round's reads returned
    // round rho = NR n + qd: record n = 3 corner + field, piece qd
    #define GEOAC_GLDS_ISSUE(rho) { \
        constexpr int n_ = (rho) / NR, qd_ = (rho) % NR, cn_ = n_ / 3, f_ = n_ % 3; \
        if(qd_ == 0 && f_ == 0) GEOAC_GLDS_BASES(cn_) \
        glds_round(tabb + (size_t)f_ * fstride + 64 * qd_, gb[0], gb[1], gb[2], gb[3], ring + 4096u * ((rho) % GEOAC_GLDS_RING), p0.x, p1.x, p2.x, p3.x); }
    GEOAC_GLDS_ISSUE(0) GEOAC_GLDS_ISSUE(1) GEOAC_GLDS_ISSUE(2) GEOAC_GLDS_ISSUE(3) GEOAC_GLDS_ISSUE(4)
    // carried between the pieces of a record
    double cF = 0, cFz = 0, cFzz = 0, cDxF = 0, cVxz = 0, cDxFzz = 0, cDyF = 0, cDxyF = 0, cDxyFz = 0, cDxVx = 0;
    auto step = [&](auto RHO) __attribute__((always_inline)) {
        constexpr int rho = decltype(RHO)::value;
        constexpr int n = rho / NR, qd = rho % NR, cn = n / 3, f = n % 3, a = cn >> 1, b = cn & 1;
        // wait until round rho has landed (rounds up to rho + 3 have been issued: iteration i >= 1 issues round i + 4), read it, and refill
        // the slot that was read ONE ITERATION AGO with round rho + 4
        constexpr int issued = rho == 0 ? 4 : (rho + 3 < NROUND - 1 ? rho + 3 : NROUND - 1);
        glds_wait<4 * (issued - rho)>();
        constexpr int so = 4096 * (rho % GEOAC_GLDS_RING);
        const geoac_d2 q0 = *(const geoac_d2*)(ra0 + so), q1 = *(const geoac_d2*)(ra1 + so), q2 = *(const geoac_d2*)(ra2 + so), q3 = *(const geoac_d2*)(ra3 + so);
        if constexpr (rho >= 1 && rho + 4 < NROUND) GEOAC_GLDS_ISSUE(rho + 4)
        p0 = q0; p1 = q1; p2 = q2; p3 = q3;
        const Cub c0 = Cub{ q0.x, q0.y, q1.x, q1.y }, c1 = Cub{ q2.x, q2.y, q3.x, q3.y };
        CornerW w = corner_weights<ORDER2, ORDER2>(hx, hy, a, b);
        const double Wq = w.W[2] * (GLB ? dys : dxs);                                 // Cartesian: Q11 row of the f_zz patch (y row scaled by dx)
        w.W[1] *= dxs; w.W[2] *= dys; w.W[3] *= dxy;
        if(ORDER2){ w.D[1] *= dxs; w.D[2] *= dys; w.D[3] *= dxy; w.E[1] *= dxs; w.E[2] *= dys; w.E[3] *= dxy; }
        double* o = M[f];
        if constexpr (qd == 0){                                                       // F, DxF
            cF = cub_val(c0, t, t6); cDxF = cub_val(c1, t, t6);
            cFz = cub_d1(c0, t, th);
            if(!GLB) cVxz = cub_d1(c1, t, th);                                        // (packed table: V_x = D_x F)
            if(ORDER2){ cFzz = cub_d2(c0, t); cDxFzz = cub_d2(c1, t); }
        } else if constexpr (qd == 1){                                                // DyF, DxyF
            const double DyF = cub_val(c0, t, t6), DxyF = cub_val(c1, t, t6);
            const double DxyFz = cub_d1(c1, t, th);
            o[0] = dot4(w.W, cF, cDxF, DyF, DxyF, o[0]);
            if(ORDER2){
                const double DyFzz = cub_d2(c0, t), DxyFzz = cub_d2(c1, t);
                o[6] = __builtin_fma(w.W[0], cFzz, __builtin_fma(w.W[1], cDxFzz, __builtin_fma(Wq, DyFzz, __builtin_fma(w.W[3], DxyFzz, o[6]))));
            }
            if(!GLB){
                const double Vyz = cub_d1(c0, t, th);                                 // (V_y = D_y F)
                o[3] = dot4(w.W, cFz, cVxz, Vyz, DxyFz, o[3]);
                if(ORDER2){
                    o[8] = dot4(w.D, cFz, cVxz, Vyz, DxyFz, o[8]);
                    o[9] = dot4(w.E, cFz, cVxz, Vyz, DxyFz, o[9]);
                }
            }
            cDyF = DyF; cDxyF = DxyF; cDxyFz = DxyFz;
        } else if constexpr (!GLB && qd == 2){                                        // DxVx, DxyVx
            const double DxVx = cub_val(c0, t, t6), DxyVx = cub_val(c1, t, t6);
            o[1] = dot4(w.W, cDxF, DxVx, cDxyF, DxyVx, o[1]);
            if(ORDER2){
                o[4] = dot4(w.D, cDxF, DxVx, cDxyF, DxyVx, o[4]);
                o[7] = dot4(w.E, cDxF, DxVx, cDxyF, DxyVx, o[7]);
            }
        } else if constexpr (GLB && qd == 2){                                         // Vx, DxVx
            cVxz = cub_d1(c0, t, th); cDxVx = cub_val(c1, t, t6);
        } else if constexpr (GLB && qd == 3){                                         // DxyVx, Vy
            const double DxyVx = cub_val(c0, t, t6), Vyz = cub_d1(c1, t, th);
            o[1] = dot4(w.W, cDxF, cDxVx, cDxyF, DxyVx, o[1]);
            o[3] = dot4(w.W, cFz, cVxz, Vyz, cDxyFz, o[3]);
            if(ORDER2){
                o[4] = dot4(w.D, cDxF, cDxVx, cDxyF, DxyVx, o[4]);
                o[7] = dot4(w.E, cDxF, cDxVx, cDxyF, DxyVx, o[7]);
                o[8] = dot4(w.D, cFz, cVxz, Vyz, cDxyFz, o[8]);
                o[9] = dot4(w.E, cFz, cVxz, Vyz, cDxyFz, o[9]);
            }
        } else {                                                                      // DyVy, DxyVy
            const double DyVy = cub_val(c0, t, t6), DxyVy = cub_val(c1, t, t6);
            o[2] = dot4(w.W, cDyF, cDxyF, DyVy, DxyVy, o[2]);
            if(ORDER2) o[5] = dot4(w.E, cDyF, cDxyF, DyVy, DxyVy, o[5]);
        }
    };
    geoac_static_for(step, std::make_integer_sequence<int, NROUND>{});
    #undef GEOAC_GLDS_ISSUE
    #undef GEOAC_GLDS_BASES
    if(ORDER2 && !GLB){                                                               // spherical set: left in scaled coordinates (Q12c)
        const double idxs = L.idxs, idys = L.idys;
        #pragma unroll
        for(int f = 0; f < 3; f++){ M[f][4] *= idxs; M[f][8] *= idxs; M[f][7] *= idys; M[f][5] *= idys; M[f][9] *= idys; }
    }
}

// ---- per-lane record cache (four lanes per ray, small fans: eigenray rounds, -interactive) ---------------------------------------
// A small fan leaves most of the chip idle and lasts as long as its longest ray: what counts is the latency of one RK4 stage, and that
// was two dependent trips to memory (z nodes, then the table records) in front of the arithmetic.  A ray stays in one cell and one
// vertical segment for tens of stages, so each lane keeps the three records (T, u, v) of ITS corner in LDS - 976 B per lane (960 B +
// 16 B pad: the ds_read_b128 / ds_write_b128 of a wave are bank-conflict free), filled from the table only when (segment, node)
// changes - and the kernel keeps a copy of the z nodes behind it.  One wave per CU (92 KB of LDS): for fans of at most 256 waves.
#define GEOAC_CACHE_SLOT 976
#ifndef GEOAC_CACHE_LDSOFF
#define GEOAC_CACHE_LDSOFF 1          // record-cache kernels read their records by explicit LDS offset (laundered per stage)
#endif
#define GEOAC_CACHE_BYTES (64 * GEOAC_CACHE_SLOT)
template <bool GLB>
DEVINL const double* grid_cache_fill(const GeoacDevParams& P, const GridLoc& L, int cq, char* cache, int* ckey){
    constexpr int RB = GRec<GLB>::N * (int)sizeof(double), NCH = RB / 16;             // bytes and 16-byte chunks per record
    const unsigned lane = threadIdx.x & 63u;
    const unsigned nn = (unsigned)(P.gnx * P.gny);
    const int key = L.kz * (int)nn + corner_node(L, cq >> 1, cq & 1);
    char* mine = cache + lane * GEOAC_CACHE_SLOT;
    if(key != *ckey){                                              // (the four lanes of a ray change cell / segment together)
        const size_t fstride = (size_t)P.nseg * nn * RB;
        const char* __restrict__ src = (const char*)P.gtab + (size_t)key * RB;
        geoac_d2 v[NCH];
        if constexpr (GLB){
            #pragma unroll
            for(int f = 0; f < 3; f++){
                #pragma unroll
                for(int j = 0; j < NCH; j++) v[j] = *(const geoac_d2*)(src + f * fstride + 16 * j);
                #pragma unroll
                for(int j = 0; j < NCH; j++) *(geoac_d2*)(mine + f * RB + 16 * j) = v[j];
            }
        } else {
            #pragma unroll 1                                       // (Cartesian kernel: unrolled, the compiler spills 189 registers around it)
            for(int f = 0; f < 3; f++){
                #pragma unroll
                for(int j = 0; j < NCH; j++) v[j] = *(const geoac_d2*)(src + f * fstride + 16 * j);
                #pragma unroll
                for(int j = 0; j < NCH; j++) *(geoac_d2*)(mine + f * RB + 16 * j) = v[j];
            }
        }
        *ckey = key;
    }
    return (const double*)mine;
}

// sixteen lanes per ray: the lane keeps ONE record - field `field` of its corner - at the head of its slot
template <bool GLB>
DEVINL const double* grid_cache_fill_one(const GeoacDevParams& P, const GridLoc& L, int cq, int field, char* cache, int* ckey){
    constexpr int RB = GRec<GLB>::N * (int)sizeof(double), NCH = RB / 16;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned nn = (unsigned)(P.gnx * P.gny);
    const int key = L.kz * (int)nn + corner_node(L, cq >> 1, cq & 1);
    char* mine = cache + lane * GEOAC_CACHE_SLOT;
    if(key != *ckey){                                              // (the sixteen lanes of a ray change cell / segment together)
        const size_t fstride = (size_t)P.nseg * nn * RB;
        const char* __restrict__ src = (const char*)P.gtab + (size_t)key * RB + (size_t)field * fstride;
        geoac_d2 v[NCH];
        #pragma unroll
        for(int j = 0; j < NCH; j++) v[j] = *(const geoac_d2*)(src + 16 * j);
        #pragma unroll
        for(int j = 0; j < NCH; j++) *(geoac_d2*)(mine + 16 * j) = v[j];
        *ckey = key;
    }
    return (const double*)mine;
}
#define GEOAC_HEX_XCHG 336            // byte offset of a ray's exchange area in the slot of its first lane (behind that lane's 320-byte record)

// Eval_Spline_f (:806-863): scalar value, y rows scaled by dx_scalar (Q11; the spherical twin :755-807 uses dp_scalar)
template <bool GLB>
DEVINL double grid_eval_f(const GeoacDevParams& P, int field, const GridLoc& L){
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    const double t = L.t, t6 = t * (1.0 / 6.0);
    const double* __restrict__ base = grid_rec<GLB>(P, field, L.kz, 0);
    const int stride = field < 3 ? GRec<GLB>::N : GEOAC_GREC_RHO;
    double v = 0.0;
    #pragma unroll 2
    for(int cn = 0; cn < 4; cn++){
        {
            const int a = cn >> 1, b = cn & 1;
            const double* __restrict__ r = base + (size_t)corner_node(L, a, b) * stride;
            Cub c[4];
            #pragma unroll
            for(int i = 0; i < 4; i++) c[i] = load_cubic(r + 4 * i);
            CornerW w = corner_weights<false, false>(hx, hy, a, b);
            w.W[1] *= L.dxs; w.W[2] *= (GLB ? L.dys : L.dxs); w.W[3] *= L.dxs * L.dys;
            v = dot4(w.W, cub_val(c[GC_F], t, t6), cub_val(c[GC_DXF], t, t6), cub_val(c[GC_DYF], t, t6), cub_val(c[GC_DXYF], t, t6), v);
        }
    }
    return v;
}
// the same value from 4 x 128 B of half-records (corners n00, n01, n10, n11) staged in LDS (k_postpass)
template <bool GLB>
DEVINL double grid_eval_f_slot(const GridLoc& L, const char* rec4){
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    const double t = L.t, t6 = t * (1.0 / 6.0);
    double v = 0.0;
    #pragma unroll 2
    for(int cn = 0; cn < 4; cn++){
        {
            const int a = cn >> 1, b = cn & 1;
            const double* __restrict__ r = (const double*)(rec4 + 128 * cn);
            Cub c[4];
            #pragma unroll
            for(int i = 0; i < 4; i++) c[i] = load_cubic(r + 4 * i);
            CornerW w = corner_weights<false, false>(hx, hy, a, b);
            w.W[1] *= L.dxs; w.W[2] *= (GLB ? L.dys : L.dxs); w.W[3] *= L.dxs * L.dys;
            v = dot4(w.W, cub_val(c[GC_F], t, t6), cub_val(c[GC_DXF], t, t6), cub_val(c[GC_DYF], t, t6), cub_val(c[GC_DXYF], t, t6), v);
        }
    }
    return v;
}
// Eval_Spline_df(.., index = 2, ..) (:920-939): df/dz patch, y rows scaled by dx_scalar (Q11); T, u, v only
// (spherical twin: Eval_Spline_df(.., index = 0, ..), :823-842, y rows scaled by dp_scalar)
template <bool GLB>
DEVINL double grid_eval_dfdz(const GeoacDevParams& P, int field, const GridLoc& L){
    const Herm hx = hermite(L.xs), hy = hermite(L.ys);
    const double t = L.t, th = 0.5 * t;
    const double* __restrict__ base = grid_rec<GLB>(P, field, L.kz, 0);
    double v = 0.0;
    #pragma unroll 1
    for(int cn = 0; cn < 4; cn++){
        {
            const int a = cn >> 1, b = cn & 1;
            const double* __restrict__ r = base + (size_t)corner_node(L, a, b) * GRec<GLB>::N;
            const Cub cF = load_cubic(r + 4 * GC_F), cXY = load_cubic(r + 4 * GC_DXYF), cX = load_cubic(r + 4 * GRec<GLB>::VX), cY = load_cubic(r + 4 * GRec<GLB>::VY);
            CornerW w = corner_weights<false, false>(hx, hy, a, b);
            w.W[1] *= L.dxs; w.W[2] *= (GLB ? L.dys : L.dxs); w.W[3] *= L.dxs * L.dys;
            v = dot4(w.W, cub_d1(cF, t, th), cub_d1(cX, t, th), cub_d1(cY, t, th), cub_d1(cXY, t, th), v);
        }
    }
    return v;
}

// scalar medium at a point (c(), u(), v(), rho() of G2S_MultiDimSpline3D.cpp:1633-1743 / G2S_GlobalMultiDimSpline3D.cpp:1502-1611,
// inputs clamped).  Arguments in table order: (x, y, z) Cartesian, (lat, lon, r) spherical; d*z = d/dz resp. d/dr.
struct Medium3 { double c, u, v, rho, dcz, duz, dvz; };
template <bool WANT_RHO, bool WANT_DZ, bool GLB = false, bool WANT_UV = true>
DEVINL Medium3 medium3_at(const GeoacDevParams& P, double x, double y, double z){
    double xe = clampd(x, P.gx[0], P.gx[P.gnx - 1]), ye = clampd(y, P.gy[0], P.gy[P.gny - 1]), ze = clampd(z, P.x_min, P.x_max);
    GridLoc L; grid_locate(P, xe, ye, ze, -1, L);
    Medium3 m;
    m.c = sqrt(kGamR * grid_eval_f<GLB>(P, 0, L));
    m.u = WANT_UV ? grid_eval_f<GLB>(P, 1, L) : 0.0;
    m.v = WANT_UV ? grid_eval_f<GLB>(P, 2, L) : 0.0;
    m.rho = WANT_RHO ? grid_eval_f<GLB>(P, 3, L) : 0.0;
    if(WANT_DZ){
        m.dcz = kGamR / (2.0 * m.c) * grid_eval_dfdz<GLB>(P, 0, L);
        m.duz = grid_eval_dfdz<GLB>(P, 1, L);
        m.dvz = grid_eval_dfdz<GLB>(P, 2, L);
    } else { m.dcz = m.duz = m.dvz = 0.0; }
    return m;
}

// fused GeoAc_UpdateSources + GeoAc_EvalSrcEq of the range-dependent Cartesian set (EquationSets.3DRngDep.cpp:218-393)
// y: x, y, z, nu_x, nu_y, nu_z | X_th(3), mu_th(3) | X_ph(3), mu_ph(3)
// NSYS = 1 (the eight-lane kernel, Eq3DRngDepOct): y = base ray | ONE launch-angle system, the lane's own
template <bool AMP, int NL = 1, bool COOP = false, bool CACHE = false, int NSYS = 2>
DEVINL void rngdep_rhs(const GeoacDevParams& P, int& kz, const double* y, double* dy, int cq = 0, char* ldsw = nullptr, int* ckey = nullptr, int* kxy = nullptr, double* cell = nullptr){
    const double xe = clampd(y[0], P.g_lo[0], P.g_hi[0]), ye = clampd(y[1], P.g_lo[1], P.g_hi[1]), ze = clampd(y[2], P.x_min, P.x_max);
    GridLoc L; grid_locate(P, xe, ye, ze, kz, L, CACHE ? (const double*)(ldsw + GEOAC_CACHE_BYTES) : nullptr, kxy, cell);
    kz = L.kz;
    double M[3][10];                                               // T, u, v and their derivatives
    // three copies of the evaluator: M[][] stays in registers (rolled, the dynamic index f put it in scratch: 240 B written and read back per stage)
#ifdef GEOAC_KSTAT
    if(COOP){   // diagnostic build: histogram of the number of DISTINCT (segment, cell) keys among the live lanes of a wave-stage
        const bool live = (*ckey & 1) != 0;
        const unsigned key = (unsigned)L.kz * 4096u + (unsigned)L.n00;
        // ... and how often a lane's own key differs from its key of the stage before (what a per-lane record cache would have to refetch)
        const bool moved = live && ((unsigned)(*ckey >> 1) != key + 1u);
        *ckey = (int)(((key + 1u) << 1) | (live ? 1u : 0u));
        const unsigned long long nmov = __popcll(__ballot(moved)), nliv = __popcll(__ballot(live));
        unsigned long long rem = __ballot(live); int K = 0;
        while(rem){ const int l = __ffsll((long long)rem) - 1; const unsigned k0 = (unsigned)__shfl((int)key, l); rem &= ~__ballot(live && key == k0); K++; }
        int kzmin = live ? L.kz : 1 << 30, kzmax = live ? L.kz : -1;
        for(int o = 32; o > 0; o >>= 1){ kzmin = min(kzmin, __shfl_xor(kzmin, o)); kzmax = max(kzmax, __shfl_xor(kzmax, o)); }
        if((threadIdx.x & 63) == 0 && K > 0){
            const int bin = K == 1 ? 0 : K == 2 ? 1 : K <= 4 ? 2 : K <= 6 ? 3 : K <= 8 ? 4 : K <= 12 ? 5 : K <= 16 ? 6 : 7;
            atomicAdd(&P.counters[16 + bin], 1ull); atomicAdd(&P.counters[24], (unsigned long long)K); atomicAdd(&P.counters[25], (unsigned long long)(kzmax - kzmin + 1));
            atomicAdd(&P.counters[26], nmov); atomicAdd(&P.counters[27], nliv);
        }
    }
#endif
    if(COOP){
        if constexpr (GRec<false>::PACKED && GEOAC_COOP_GLDS) grid_eval3_glds<AMP>(P, L, M, ldsw);
        else if constexpr (GRec<false>::PACKED) grid_eval3_coop8<AMP>(P, L, M, ldsw);
        else grid_eval3_coop<AMP, false>(P, L, M, ldsw);
    } else if(CACHE){
        const double* rec = grid_cache_fill<false>(P, L, cq, ldsw, ckey);
        if(GEOAC_CACHE_LDSOFF){
            unsigned off = (unsigned)(size_t)(geoac_lds_char*)rec;
            asm volatile("" : "+v"(off));
            #pragma unroll
            for(int f = 0; f < 3; f++) grid_eval_all<AMP, false, NL, true>(P, f, L, M[f], cq, rec, off + f * (unsigned)(GRec<false>::N * sizeof(double)));
        } else {
            #pragma unroll
            for(int f = 0; f < 3; f++) grid_eval_all<AMP, false, NL>(P, f, L, M[f], cq, rec + f * GRec<false>::N);
        }
    } else {
        #pragma unroll
        for(int f = 0; f < 3; f++) grid_eval_all<AMP, false, NL>(P, f, L, M[f], cq);
    }
    const double* T = M[0]; const double* U = M[1]; const double* V = M[2];
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
        for(int a = 0; a < NSYS; a++){
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

// fused GeoAc_UpdateSources + GeoAc_EvalSrcEq of the range-dependent spherical set (EquationSets.GlobalRngDep.cpp:226-458):
// the algebra of global_rhs (geoac_kernels.hip) with the full gradient and second-derivative matrices of c, u, v (w = 0).
// y: r, lat, lon, nu_r, nu_t, nu_p | R_lt(3), mu_lt(3) | R_lp(3), mu_lp(3);  sth/cth = sin/cos(lat) from the caller.
// NSYS = 1 (the eight-lane kernel, EqGlobalRngDepOct): y = base ray | ONE launch-angle system, the lane's own
// FSPLIT (sixteen lanes per ray: EqGlobalRngDepHex): the quad (lane >> 2) & 3 of a ray's sixteen lanes evaluates ONE field (T, u, v; the fourth quad
// repeats T and keeps its result to itself) - a third of the table evaluation per lane - and the three fields' ten values each change hands
// through 240 bytes of LDS in the slot of the ray's first lane.  Per field the same corner sums in the same order as the four- and eight-lane
// kernels: the same bits.
template <bool AMP, int NL = 1, bool COOP = false, bool CACHE = false, int NSYS = 2, bool FSPLIT = false>
DEVINL void globalrd_rhs(const GeoacDevParams& P, int& kz, const double* y, double sth, double cth, double* dy, int cq = 0, char* ldsw = nullptr, int* ckey = nullptr, int* kxy = nullptr, double* cell = nullptr){
    const double r = y[0];
    const double te = clampd(y[1], P.g_lo[0], P.g_hi[0]), pe = clampd(y[2], P.g_lo[1], P.g_hi[1]), re = clampd(r, P.x_min, P.x_max);
    GridLoc L; grid_locate(P, te, pe, re, kz, L, CACHE ? (const double*)(ldsw + GEOAC_CACHE_BYTES) : nullptr, kxy, cell);
    kz = L.kz;
    double M[3][10];                                               // table order: f, f_t, f_p, f_r, f_tt, f_pp, f_rr, f_tp, f_tr, f_pr
    if(COOP){
        if constexpr (GEOAC_COOP_GLDS != 0) grid_eval3_glds<AMP, true>(P, L, M, ldsw);
        else grid_eval3_coop<AMP, true>(P, L, M, ldsw);
    }
    else if(CACHE && FSPLIT){
        const unsigned lane = threadIdx.x & 63u;
        const int fq = (int)((lane >> 2) & 3u), fm = fq < 3 ? fq : 0;
        const double* rec = grid_cache_fill_one<true>(P, L, cq, fm, ldsw, ckey);
        unsigned off = (unsigned)(size_t)(geoac_lds_char*)rec;      // (explicit LDS reads, offset laundered per stage: as in the eight-lane kernel)
        asm volatile("" : "+v"(off));
        double o[10];
        grid_eval_all<AMP, true, NL, true>(P, fm, L, o, cq, rec, off);
        // the quad's totals (identical in its four lanes) to the ray's exchange area, then every lane reads all three fields
        char* const xa = ldsw + (lane & ~15u) * GEOAC_CACHE_SLOT + GEOAC_HEX_XCHG + 16u * (lane >> 4);
        constexpr int NX = AMP ? 5 : 2;                            // 16-byte pieces per field: ten values with the second derivatives, four without
        if(cq == 0 && fq < 3){
            #pragma unroll
            for(int i = 0; i < NX; i++){ geoac_d2 w; w.x = o[2 * i]; w.y = o[2 * i + 1]; *(geoac_d2*)(xa + 80 * fq + 16 * i) = w; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        #pragma unroll
        for(int f = 0; f < 3; f++){
            #pragma unroll
            for(int i = 0; i < NX; i++){ const geoac_d2 w = *(const geoac_d2*)(xa + 80 * f + 16 * i); M[f][2 * i] = w.x; M[f][2 * i + 1] = w.y; }
            #pragma unroll
            for(int i = 2 * NX; i < 10; i++) M[f][i] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");              // the next stage's stores stay behind these reads
        __builtin_amdgcn_wave_barrier();
    }
    else if(CACHE){
        const double* rec = grid_cache_fill<true>(P, L, cq, ldsw, ckey);
        if(NSYS == 1 || GEOAC_CACHE_LDSOFF){                       // eight-lane kernel: keep the record reads in the stage loop (hipcc otherwise parks the
            unsigned off = (unsigned)(size_t)(geoac_lds_char*)rec;  // three records in AGPRs across the stages: 230 v_accvgpr moves per stage instead of 60
            asm volatile("" : "+v"(off));                          // ds_read_b128): the LDS offset is laundered and the reads are explicit LDS reads
            #pragma unroll
            for(int f = 0; f < 3; f++) grid_eval_all<AMP, true, NL, true>(P, f, L, M[f], cq, rec, off + f * (unsigned)(GRec<true>::N * sizeof(double)));
        } else {
            #pragma unroll
            for(int f = 0; f < 3; f++) grid_eval_all<AMP, true, NL>(P, f, L, M[f], cq, rec + f * GRec<true>::N);
        }
    } else {
        #pragma unroll
        for(int f = 0; f < 3; f++) grid_eval_all<AMP, true, NL>(P, f, L, M[f], cq);
    }
    const double* T = M[0]; const double* U = M[1]; const double* V = M[2];
    // first derivatives in equation order (r, t, p) = table entries 3, 1, 2
    const int fi[3] = { 3, 1, 2 };
    const double n0 = y[3], n1 = y[4], n2 = y[5];
    const double u = U[0], v = V[0];
    const double qT = kGamR * T[0];
    const double ic = frsq(qT);
    const double c  = qT * ic;
    const double hc = (0.5 * kGamR) * ic;
    double dc[3], du[3], dv[3];
    #pragma unroll
    for(int i = 0; i < 3; i++){ dc[i] = hc * T[fi[i]]; du[i] = U[fi[i]]; dv[i] = V[fi[i]]; }
    const double nn  = __builtin_fma(n0, n0, __builtin_fma(n1, n1, n2 * n2));
    const double inm = frsq(nn);
    const double numag = nn * inm;
    const double cn  = c * inm;
    const double cg0 = cn * n0;
    const double cg1 = __builtin_fma(cn, n1, v);
    const double cg2 = __builtin_fma(cn, n2, u);
    const double icg = frsq(__builtin_fma(cg0, cg0, __builtin_fma(cg1, cg1, cg2 * cg2)));
    const double ir  = frcp(r);
    const double ico = frcp(cth);
    const double tn  = sth * ico;
    const double u0 = cg0 * icg, u1 = cg1 * icg, u2 = cg2 * icg;
    const double G[3] = { 1.0, ir, ir * ico };                       // GeoCoeff (:258-260)
    // GeoTerms (:262-268) with c_g substituted (see global_rhs)
    const double nc12 = __builtin_fma(n1, cg1, n2 * cg2);
    const double ncs  = __builtin_fma(n0, cth, n1 * sth);
    const double n2cg2 = n2 * cg2;
    double Tg[3], H[3];
    Tg[0] = ir * nc12;
    Tg[1] = __builtin_fma(n2cg2, tn, -(cn * n0) * n1);
    Tg[2] = -n2 * __builtin_fma(cn, ncs, v * sth);
    #pragma unroll
    for(int i = 0; i < 3; i++) H[i] = __builtin_fma(numag, dc[i], __builtin_fma(n1, dv[i], n2 * du[i]));
    dy[0] = u0;
    dy[1] = G[1] * u1;
    dy[2] = G[2] * u2;
    #pragma unroll
    for(int i = 0; i < 3; i++) dy[3 + i] = -(G[i] * icg) * (H[i] + Tg[i]);

    if(AMP){
        // second derivatives in equation order: [i][m] over (r, t, p) from table entries rr = 6, tt = 4, pp = 5, rt = 8, rp = 9, tp = 7
        const int ij[3][3] = { {6, 8, 9}, {8, 4, 7}, {9, 7, 5} };
        const double hc3 = (0.25 * kGamR * kGamR) * (ic * ic * ic);
        const double ir2 = ir * ir, ico2 = ico * ico;
        const double cnn1 = cn * n1, cnn2 = cn * n2;
        #pragma unroll
        for(int q = 0; q < NSYS; q++){
            const double R[3] = { y[6 + 6 * q], y[7 + 6 * q], y[8 + 6 * q] };
            const double m0 = y[9 + 6 * q], m1 = y[10 + 6 * q], m2 = y[11 + 6 * q];
            double dca = 0.0, dua = 0.0, dva = 0.0;
            #pragma unroll
            for(int n = 0; n < 3; n++){ dca = __builtin_fma(R[n], dc[n], dca); dua = __builtin_fma(R[n], du[n], dua); dva = __builtin_fma(R[n], dv[n], dva); }
            const double dnu = __builtin_fma(n0, m0, __builtin_fma(n1, m1, n2 * m2)) * inm;
            const double al  = inm * __builtin_fma(-cn, dnu, dca);
            const double a1  = __builtin_fma(n1, al, cn * m1);
            const double a2  = __builtin_fma(n2, al, cn * m2);
            const double dcg0 = __builtin_fma(n0, al, cn * m0);
            const double dcg1 = a1 + dva;
            const double dcg2 = a2 + dua;
            const double e   = icg * __builtin_fma(u0, dcg0, __builtin_fma(u1, dcg1, u2 * dcg2));
            const double w0 = __builtin_fma(icg, dcg0, -u0 * e);
            const double w1 = __builtin_fma(icg, dcg1, -u1 * e);
            const double w2 = __builtin_fma(icg, dcg2, -u2 * e);
            double dG[3];
            dG[0] = 0.0;
            dG[1] = -R[0] * ir2;
            dG[2] = G[2] * __builtin_fma(tn, R[1], -R[0] * ir);
            const double s22 = __builtin_fma(m2, cg2, n2 * dcg2);
            double dT[3];
            dT[0] = __builtin_fma(dG[1], nc12, ir * __builtin_fma(m1, cg1, __builtin_fma(n1, dcg1, s22)));
            dT[1] = __builtin_fma(-cnn1, m0, __builtin_fma(-n0, a1, __builtin_fma(tn, s22, (n2cg2 * R[1]) * ico2)));
            const double dncs = __builtin_fma(m0, cth, __builtin_fma(m1, sth, R[1] * __builtin_fma(n1, cth, -n0 * sth)));
            dT[2] = -__builtin_fma(__builtin_fma(m2, v, n2 * dva), sth,
                                   __builtin_fma((n2 * v) * R[1], cth, __builtin_fma(a2, ncs, cnn2 * dncs)));
            dy[6 + 6 * q] = w0;
            dy[7 + 6 * q] = __builtin_fma(dG[1], u1, G[1] * w1);
            dy[8 + 6 * q] = __builtin_fma(dG[2], u2, G[2] * w2);
            #pragma unroll
            for(int i = 0; i < 3; i++){
                double ddc = 0.0, ddu = 0.0, ddv = 0.0;
                #pragma unroll
                for(int mm = 0; mm < 3; mm++){
                    ddc = __builtin_fma(R[mm], __builtin_fma(hc, T[ij[i][mm]], -hc3 * (T[fi[i]] * T[fi[mm]])), ddc);
                    ddu = __builtin_fma(R[mm], U[ij[i][mm]], ddu);
                    ddv = __builtin_fma(R[mm], V[ij[i][mm]], ddv);
                }
                const double K = __builtin_fma(numag, ddc, __builtin_fma(n1, ddv, n2 * ddu));
                // (:437-445)  -dG/|cg| (H + T) + G/|cg|^2 d|cg| H - G/|cg| (d|nu| dc + mu.(dv, du) + K + dT)
                dy[9 + 6 * q + i] = icg * (-dG[i] * (H[i] + Tg[i])
                                           + G[i] * (e * H[i] - (__builtin_fma(dnu, dc[i], __builtin_fma(m1, dv[i], __builtin_fma(m2, du[i], K))) + dT[i])));
            }
        }
    }
}

#endif
