/* geoac_host.h - host-side (CPU, set-up only) helpers of libgeoac_hip.so: the pieces of GeoAc that run
 * once per job and feed the GPU path.  They replace
 *   file_length / Load_G2S / Spline_Single_G2S     Code/Atmo/G2S_Spline1D.cpp:55-142,293-309
 *                                                  Code/Atmo/G2S_GlobalSpline1D.cpp:58-152,305-320
 *   Set_Slopes                                     Code/Atmo/G2S_Spline1D.cpp:161-196
 *   the launch-angle loop headers                  Code/GeoAcGlobal_main.cpp:241-242 (and twins)
 * No ray is ever integrated on the host.
 */
#ifndef GEOAC_HOST_H_
#define GEOAC_HOST_H_

#ifdef __cplusplus
extern "C" {
#endif

/* number of profile rows the reference would read from `file` (= number of '\n'), <0 on error */
int  geoac_met_rows(const char* file);

/* read a .met profile (format "zTuvdp" or "zuvwTdp").  Outputs (each `cap` >= rows doubles):
 * x = altitude [km] (+ r_earth for eqset GEOAC_EQ_GLOBAL*), T [K], u,v [km/s] with the reference's
 * ground taper (z_grnd = 0 at load time: the mains parse z_grnd= after loading), rho.  Returns rows. */
int  geoac_met_load(const char* file, const char* format, int eqset, int cap,
                    double* x, double* T, double* u, double* v, double* rho);

/* the same with the taper centred on z_grnd: GeoAcGlobal -interactive loads the profile a second time AFTER parsing z_grnd=
 * (GeoAcGlobal_main.cpp:367), the only stratified entry point where z_grnd reaches the taper (Q9) */
int  geoac_met_load_zg(const char* file, const char* format, int eqset, double z_grnd, int cap,
                       double* x, double* T, double* u, double* v, double* rho);
int  geoac_met_from_columns_zg(int eqset, double z_grnd, int n, const double* z, const double* T, const double* u_ms,
                               const double* v_ms, const double* rho_in,
                               double* x, double* T_out, double* u, double* v, double* rho);

/* the same transformation applied to columns already in memory (z [km], T [K], u,v [m/s], rho) */
int  geoac_met_from_columns(int eqset, int n, const double* z, const double* T, const double* u_ms,
                            const double* v_ms, const double* rho_in,
                            double* x, double* T_out, double* u, double* v, double* rho);

/* range-dependent grid of profiles <prefix><n>.met (n = ix*ny + iy, G2S_MultiDimSpline3D.cpp:154) with node coordinates in
 * locx / locy (one value per line).  geoac_grid_dims returns the counts the reference would use (newline counts);
 * geoac_grid_load fills x[nx], y[ny], z[nz] and the fields [nx][ny][nz] with the reference's taper (width 0.05 km around
 * z_grnd: the RngDep mains parse z_grnd= BEFORE loading, :167-168) and m/s -> km/s. */
int  geoac_grid_dims(const char* prefix, const char* locx, const char* locy, int* nx, int* ny, int* nz);
int  geoac_grid_load(const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd,
                     int nx, int ny, int nz, double* x, double* y, double* z,
                     double* T, double* u, double* v, double* rho);

/* the same for either range-dependent set.  GEOAC_EQ_GLOBAL_RNGDEP: locx / locy hold latitudes / longitudes in degrees (returned in
 * radians), z comes back as geocentric radius (+6370 km), taper width 0.2 km (G2S_GlobalMultiDimSpline3D.cpp:138-195) */
int  geoac_grid_load_eq(int eqset, const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd,
                        int nx, int ny, int nz, double* x, double* y, double* z,
                        double* T, double* u, double* v, double* rho);

/* table of the range-dependent Cartesian interpolant as the kernels read it (layout: geoac_amd/csrc/geoac_rngdep.h).  Per
 * (field, kz, node) the vertical cubics (c0, c1, 2 c2, 6 c3) of F, DxF, DyF, DxyF, Vx, DxVx, DxyVx, Vy, DyVy, DxyVy for T, u, v
 * (40 doubles) and F, DxF, DyF, DxyF for rho (16 doubles): V0 = S_f, Vx = S_fx, Vy = S_fy are the reference's three vertical
 * natural splines per node (Set_Slopes_Multi, G2S_MultiDimSpline3D.cpp:306-425); D* are its evaluation-time finite differences
 * (BiCubic_Deriv_*, :568-800), linear in the coefficients and centred at the node, taken once on the coefficients.
 * Fields are [nx][ny][nz] as geoac_grid_load returns them. */
size_t geoac_grid_table_size(int nx, int ny, int nz);
int    geoac_grid_table(int nx, int ny, int nz, const double* x, const double* y, const double* z,
                        const double* T, const double* u, const double* v, const double* rho, double* tab);
/* scalar evaluation of field 0..3 (T, u, v, rho) at a point from that table, as the reference's Eval_Spline_f does (inputs
 * clamped to the grid; y rows scaled by the x cell size, Q11).  Reporting only (atmo.dat): rays never come through here. */
double geoac_grid_eval(int nx, int ny, int nz, const double* x, const double* y, const double* z, const double* tab,
                       int field, double xq, double yq, double zq);

/* either range-dependent set.  The spherical set (x = latitude, y = longitude [rad], z = radius) keeps the reference's quirks Q12:
 * the S_fx / S_fy slope systems use its interior right-hand side, and the Vx / Vy cubics carry the truncated z-derivative;
 * its scalar evaluator scales the y rows by the y cell size. */
int    geoac_grid_table_eq(int eqset, int nx, int ny, int nz, const double* x, const double* y, const double* z,
                           const double* T, const double* u, const double* v, const double* rho, double* tab);
double geoac_grid_eval_eq(int eqset, int nx, int ny, int nz, const double* x, const double* y, const double* z, const double* tab,
                          int field, double xq, double yq, double zq);

/* cubic of one spline segment (values f0, f1 and slopes s0, s1 at x0 < x1) in powers of t = x - x0: c = (c0, c1, c2, c3), or
 * (c0, c1, 2 c2, 6 c3) when deriv_form != 0 */
void geoac_spline_segment_cubic(double x0, double x1, double f0, double f1, double s0, double s1, double* c, int deriv_form);

/* natural cubic spline node slopes (Thomas algorithm, natural end conditions) */
void geoac_natural_spline_slopes(int n, const double* x, const double* f, double* slopes);

/* enumerate `for(phi = phi_min; phi <= phi_max; phi += phi_step) for(theta = ...)` by repeated addition,
 * phi outer.  Returns the ray count (also when cap is too small; nothing is written past cap). */
long geoac_fan_enumerate(double theta_min, double theta_max, double theta_step,
                         double phi_min, double phi_max, double phi_step,
                         long cap, double* theta_out, double* phi_out);

#ifdef __cplusplus
}
#endif
#endif
