// geoac_eigenray.cpp - batched eigenray searches on top of the ray-fan C ABI (include/geoac_eig.h): spherical sets
// (GeoAc.Eigenray.Global.cpp) and 3-D Cartesian sets (GeoAc.Eigenray.cpp).
//
// A search is a set of TASKS - resumable state machines that post rays and consume arrival records:
//   ScanChain   one per (receiver, bounce count): header, then estimate -> (refinement spawned) -> estimate -> ... over the inclination range
//   Estimator   GeoAc_EstimateEigenray: up to five inclination scans; a scan whose step does not depend on earlier arrivals (the first
//               three) is posted as ONE request and replayed in the reference's order, the adaptive ones ray by ray
//   Refiner     GeoAc_3DEigenray_LM: the damped Newton iteration on the launch angles, one ray per iteration, then the eigenray's own
//               ray with the raypath samples
// What differs between the spherical and the Cartesian mains - geometry, the derivative matrix, the wording of the log - sits in a Family.
// The scheduler (run_all) is a plain loop on the calling thread: advance every task until it waits for a ray or ends, integrate all posted
// requests of the round - grouped by (bounces, CalcAmp, output mode), one fan launch per group, the groups side by side on clones of the
// context - hand the records back, repeat.  Nothing an estimate or a refinement computes depends on another bounce count or receiver, so
// all of them share the rounds; the text each task would have written to the reference's cout is kept per task and rendered in the
// reference's order at the end.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <iomanip>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/geoac_eig.h"

namespace {

const double Pi = 3.141592653589793238462643;

struct Request {
    int bounces = 0, calc_amp = 0, mode = 0;
    std::vector<double> th, ph;            // launch inclination, azimuth from north [deg]
    std::vector<double> rec;               // [n][bounces+1][GEOAC_REC_STRIDE]
    std::vector<double> smp;               // sample rows of these rays (GEOAC_SMP_RAY = index within the request)
};

struct Eigenray { double v[GEOAC_EIG_STRIDE]; std::vector<double> smp; };

struct Geo {                               // spherical helpers of GeoAc.Eigenray.Global.cpp:25-43
    double r_earth;
    double bearing(double lat1, double long1, double lat2, double long2) const {
        double term1 = sin((long2 - long1) * Pi / 180.0);
        double term2 = cos(lat1 * Pi / 180.0) * tan(lat2 * Pi / 180.0) - sin(lat1 * Pi / 180.0) * cos((long2 - long1) * Pi / 180.0);
        return atan2(term1, term2) * 180.0 / Pi;
    }
    double gc_distance(double lat1, double long1, double lat2, double long2) const {
        double term1 = pow(sin((lat2 - lat1) * Pi / 180.0 / 2.0), 2);
        double term2 = cos(lat1 * Pi / 180.0) * cos(lat2 * Pi / 180.0) * pow(sin((long2 - long1) * Pi / 180.0 / 2.0), 2);
        return 2.0 * r_earth * asin(sqrt(term1 + term2));
    }
};

const double d_theta_big = 0.25, d_theta_small = 0.002;     // GeoAc.Eigenray.Global.cpp:21-22
double modify_d_theta(double dr, double dr_dtheta){           // :40-44
    double width = 1.0 / 2.0 * pow(dr_dtheta, 2);
    return d_theta_big - (d_theta_big - d_theta_small) * exp(-dr * dr / width);
}

// Deferred log: what a task would have written to the reference's cout, kept as operations and rendered at the end into ONE stream per
// receiver in the reference's order - the tasks of a receiver advance side by side, but a std::setprecision of an earlier block must
// still stick to the later ones, and the running eigenray number is only known once the earlier blocks are complete.
struct DLog {
    enum Kind { TEXT, NUM, INT, PREC, COUNT, INCR };
    struct Op { Kind k; std::string s; double d; long long i; };
    struct Count {};                                             // prints the number of eigenrays identified so far
    std::vector<Op> ops;
    DLog& text(const std::string& t){ if(!ops.empty() && ops.back().k == TEXT) ops.back().s += t; else ops.push_back(Op{TEXT, t, 0.0, 0}); return *this; }
    DLog& operator<<(const char* t){ return text(t); }
    DLog& operator<<(const std::string& t){ return text(t); }
    DLog& operator<<(char c){ return text(std::string(1, c)); }
    DLog& operator<<(double d){ ops.push_back(Op{NUM, "", d, 0}); return *this; }
    DLog& operator<<(int i){ ops.push_back(Op{INT, "", 0.0, i}); return *this; }
    struct Prec { int n; };                                      // the log's own setprecision (std::setprecision's return type hides its argument)
    DLog& operator<<(Prec pr){ ops.push_back(Op{PREC, "", 0.0, (long long)pr.n}); return *this; }
    DLog& operator<<(Count){ ops.push_back(Op{COUNT, "", 0.0, 0}); return *this; }
    void incr(){ ops.push_back(Op{INCR, "", 0.0, 0}); }
    void render(std::ostringstream& os, int& count) const {
        for(const Op& o : ops){
            switch(o.k){
                case TEXT: os << o.s; break;
                case NUM: os << o.d; break;
                case INT: os << o.i; break;
                case PREC: os << std::setprecision((int)o.i); break;
                case COUNT: os << count; break;
                case INCR: count++; break;
            }
        }
    }
};

// one task's output: its log and the eigenrays it identified
struct Segment { DLog log; std::vector<Eigenray> found; };

// outcome of ray i of a request: BreakCheck of the reference's leg loop, last row solution[k][*]
bool broke(const Request& rq, int i){
    const int legs = rq.bounces + 1;
    const double* R = &rq.rec[((size_t)i * legs + (legs - 1)) * GEOAC_REC_STRIDE];
    return R[GEOAC_REC_VALID] == 0.0;
}
const double* last_row(const Request& rq, int i){
    const int legs = rq.bounces + 1;
    // the row the reference reads after a break is the breaking leg's last row; only used for messages there
    int l = legs - 1;
    while(l > 0 && rq.rec[((size_t)i * legs + l) * GEOAC_REC_STRIDE + GEOAC_REC_STEPS] == 0.0) l--;
    return &rq.rec[((size_t)i * legs + l) * GEOAC_REC_STRIDE];
}

// what every task of one receiver reads
struct Site {
    int eqset = GEOAC_EQ_GLOBAL;
    int rcvr_index = 0;
    double src[3] = {0, 0, 0};             // Source_Loc of the reference: (lat, lon [deg], z) or (x, y, z) [km]
    double rcv[2] = {0, 0};
    double z_grnd = 0.0;
    geoac_eig_params prm{};
    bool verbose = false;
};

// the numbers a refinement carries from one iteration to the next (GeoAc_3DEigenray_LM's locals; the long doubles are the reference's)
struct Newton {
    double lt = 0, lp = 0;                 // launch inclination and azimuth [deg] as the routine carries them (spherical: lp = 90 - azimuth from north; Cartesian: azimuth from east)
    double dr = 0, dr_prev = 10000.0, step_scalar = 1.0;
    long double dlt = 0, dlp = 0;          // the last full step
    long double p = 0, q = 0;              // arrival: (lat, lon) [rad] or (x, y) [km]
    double nu0_xy[2] = {0, 0};             // Cartesian stratified set: horizontal slowness of the launch direction
};

// ================= what the spherical and the Cartesian mains do differently =================
struct Family {
    Site site;
    virtual ~Family(){}
    // the text around the -eig_search driver loop
    virtual void header(DLog& log, int n_bnc) const = 0;
    virtual void footer(DLog& log) const = 0;
    // ---- GeoAc_EstimateEigenray ----
    virtual double range_to_receiver() const = 0;
    virtual double first_azimuth() const = 0;                                      // the azimuth the scans start with, in the routine's own convention
    virtual double azimuth_from_north(double phi) const = 0;                       // ... as the fan ABI wants it
    virtual double azimuth_estimate(double phi) const = 0;                         // ... as the refinement wants it
    virtual bool scan_goes_on(double theta, double theta_max) const = 0;
    virtual void announce(DLog& log, double range0, double phi, double theta_min, double theta_max) const = 0;
    virtual double arrival(DLog& log, double theta, int bounces, bool left_region, const double* Rk, double range0) const = 0;   // range of one ray of a scan + its line
    virtual double azimuth_deviation(const double* Rk) const = 0;
    virtual void verdict(DLog& log, bool acceptable, double d_phi, double limit) const = 0;
    virtual double next_step(double dr, double dr_dtheta) const = 0;
    virtual void gave_up(DLog& log) const = 0;
    // ---- GeoAc_3DEigenray_LM ----
    virtual void launch_direction(Newton& N) const {}
    virtual void left_region(DLog& log) const {}
    virtual void maxed_out(DLog& log) const = 0;
    virtual double miss(DLog* log, Newton& N, const double* S, int bounces) const = 0;          // distance between arrival and receiver (+ its line of the log)
    virtual void full_step(Newton& N, const double* S) const = 0;                                // the Newton step on (lt, lp), clamped: N.dlt, N.dlp
    virtual void identified(Segment& out, const Newton& N, const Request& fin, int bounces) const = 0;   // the eigenray's record and its block of the log
};

// ---- spherical sets: GeoAc.Eigenray.Global.cpp ----
struct SphericalFamily : Family {
    Geo geo{6370.0};
    void header(DLog& log, int n_bnc) const override { log << "Searching for " << n_bnc << " bounce eigenrays." << '\n'; }      // GeoAcGlobal_main.cpp:566-580
    void footer(DLog& log) const override { log << "Identified " << DLog::Count{} << " eigenray(s)." << '\n'; }
    // GeoAc_EstimateEigenray: GeoAc.Eigenray.Global.cpp:46-136
    double range_to_receiver() const override { return geo.gc_distance(site.src[0], site.src[1], site.rcv[0], site.rcv[1]); }
    double first_azimuth() const override { return geo.bearing(site.src[0], site.src[1], site.rcv[0], site.rcv[1]); }
    double azimuth_from_north(double phi) const override { return phi; }
    double azimuth_estimate(double phi) const override { return 90.0 - phi; }
    bool scan_goes_on(double theta, double theta_max) const override { return theta < theta_max; }
    void announce(DLog& log, double range0, double phi, double theta_min, double theta_max) const override {
        log << '\t' << "Estimating eigenray angles for source-receiver separated by great circle distance " << range0 << " km, and azimuth " << phi;
        log << " degrees from N.  Inclination limits: [" << theta_min << ", " << theta_max << "]." << '\n';
    }
    double arrival(DLog& log, double theta, int bounces, bool left, const double* Rk, double range0) const override {
        const double lat_k = Rk[GEOAC_REC_STATE + 1] * 180.0 / Pi, lon_k = Rk[GEOAC_REC_STATE + 2] * 180.0 / Pi;
        const double r = left ? range0 : geo.gc_distance(site.src[0], site.src[1], lat_k, lon_k);
        if(site.verbose){
            log << '\t' << '\t' << "Ray launched at inclination=" << (theta * Pi / 180.0) * 180.0 / Pi << " degrees arrives at range " << r;
            log << " km after " << bounces << " bounces.  Exact arrival at " << lat_k << " degrees N latitude, " << lon_k << " degrees E longitude" << '\n';
        }
        return r;
    }
    double azimuth_deviation(const double* Rk) const override {
        const double lat_k = Rk[GEOAC_REC_STATE + 1] * 180.0 / Pi, lon_k = Rk[GEOAC_REC_STATE + 2] * 180.0 / Pi;
        double d_phi = geo.bearing(site.src[0], site.src[1], site.rcv[0], site.rcv[1]);
        d_phi -= geo.bearing(site.src[0], site.src[1], lat_k, lon_k);
        return d_phi;
    }
    void verdict(DLog& log, bool ok, double d_phi, double limit) const override {
        if(ok) log << '\t' << '\t' << "Azimuth deviation less than " << limit << " degrees.  Estimates acceptable." << '\n' << '\n';
        else   log << '\t' << '\t' << "Azimuth deviation greater than " << limit << " degrees.  Compensating and searching inclinations again." << '\n' << '\n';
    }
    double next_step(double dr, double dr_dtheta) const override { return modify_d_theta(dr, dr_dtheta); }
    void gave_up(DLog& log) const override { log << '\t' << '\t' << "Reached maximum inclination angle or iteration limit." << '\n' << '\n'; }
    // GeoAc_3DEigenray_LM: GeoAc.Eigenray.Global.cpp:139-319
    void maxed_out(DLog& log) const override { log << '\t' << '\t' << '\t' << "Search for exact eigenray maxed out iterations.  No eigneray idenfied." << '\n'; }
    double miss(DLog* log, Newton& N, const double* S, int) const override {
        N.p = S[1]; N.q = S[2];
        const double dr = geo.gc_distance((double)(N.p * 180.0 / Pi), (double)(N.q * 180.0 / Pi), site.rcv[0], site.rcv[1]);
        if(site.verbose && log) *log << '\t' << '\t' << "Arrival at (" << DLog::Prec{8} << (double)(N.p * 180.0 / Pi) << ", " << (double)(N.q * 180.0 / Pi) << "), distance to receiver = " << dr << " km." << '\n';
        return dr;
    }
    void full_step(Newton& N, const double* S) const override {
        const long double lat = N.p, lon = N.q;
        long double d_lat, d_lon, d_lat_dlt, d_lon_dlt, d_lat_dlp, d_lon_dlp, det;
        const double lt_lim_step = 0.2, lp_lim_step = 0.2;
        d_lat = site.rcv[0] * Pi / 180.0 - lat;
        d_lon = site.rcv[1] * Pi / 180.0 - lon;
        const double rg = geo.r_earth + site.z_grnd;
        d_lat_dlt = S[7]  - 1.0 / rg * S[4] / S[3] * S[6];
        d_lat_dlp = S[13] - 1.0 / rg * S[4] / S[3] * S[12];
        d_lon_dlt = S[8]  - 1.0 / (rg * cos(lat)) * S[5] / S[3] * S[6];
        d_lon_dlp = S[14] - 1.0 / (rg * cos(lat)) * S[5] / S[3] * S[12];
        det = d_lat_dlt * d_lon_dlp - d_lat_dlp * d_lon_dlt;
        N.dlt = (d_lon_dlp * d_lat - d_lat_dlp * d_lon) / det * 180.0 / Pi;
        N.dlp = (-d_lon_dlt * d_lat + d_lat_dlt * d_lon) / det * 180.0 / Pi;
        if(N.dlt >  lt_lim_step) N.dlt =  lt_lim_step;
        if(N.dlt < -lt_lim_step) N.dlt = -lt_lim_step;
        if(N.dlp >  lp_lim_step) N.dlp =  lp_lim_step;
        if(N.dlp < -lp_lim_step) N.dlp = -lp_lim_step;
    }
    void identified(Segment& out, const Newton& N, const Request& fin, int bnc_cnt) const override {
        // the reference re-propagates and accumulates travel time / attenuation with the raypath-writing loop (:198-238): the request was the
        // same ray again with WriteRays on (segment form of Q7, samples every 25th step)
        DLog& log = out.log;
        const double lt = N.lt, lp = N.lp;
        const double* R = &fin.rec[((size_t)bnc_cnt) * GEOAC_REC_STRIDE];
        const double* Sk = R + GEOAC_REC_STATE;
        Eigenray e; memset(e.v, 0, sizeof e.v);
        const double travel_time = R[GEOAC_REC_TTIME], attenuation = R[GEOAC_REC_ATTEN];
        // arrival inclination: -asin(c_k / c_src nu_r) (:241); the fan record of the range-dependent main carries the opposite sign (Q10)
        double arrival_incl = (site.eqset == GEOAC_EQ_GLOBAL_RNGDEP) ? -R[GEOAC_REC_INCL] : R[GEOAC_REC_INCL];
        double bearing_back = geo.bearing(site.rcv[0], site.rcv[1], site.src[0], site.src[1]);
        double back_az = 90.0 - atan2(-Sk[4], -Sk[5]) * 180.0 / Pi;
        double back_az_dev = back_az - bearing_back;
        if(back_az_dev > 180.0)  back_az_dev -= 360.0;
        if(back_az_dev < -180.0) back_az_dev += 360.0;
        e.v[GEOAC_EIG_RCVR] = site.rcvr_index; e.v[GEOAC_EIG_INDEX] = 0;  /* numbered when the receiver's segments are put in order */ e.v[GEOAC_EIG_BOUNCES] = bnc_cnt;
        e.v[GEOAC_EIG_THETA] = lt; e.v[GEOAC_EIG_PHI] = 90.0 - lp;
        e.v[GEOAC_EIG_TTIME] = travel_time;
        e.v[GEOAC_EIG_CELERITY] = geo.gc_distance(site.src[0], site.src[1], site.rcv[0], site.rcv[1]) / travel_time;
        e.v[GEOAC_EIG_AMP_DB] = 20.0 * log10(R[GEOAC_REC_AMP]);
        e.v[GEOAC_EIG_ATTEN_DB] = -attenuation;
        e.v[GEOAC_EIG_INCL] = arrival_incl; e.v[GEOAC_EIG_BEARING] = bearing_back; e.v[GEOAC_EIG_BACKAZ] = back_az; e.v[GEOAC_EIG_AZDEV] = back_az_dev;
        e.smp = fin.smp;
        e.v[GEOAC_EIG_NSMP] = (double)(e.smp.size() / GEOAC_SMP_STRIDE);
        if(site.verbose){
            log << '\t' << '\t' << "Eigenray-" << DLog::Count{} << ".  " << bnc_cnt << " bounce(s)." << '\n';
            log << '\t' << '\t' << '\t' << "theta, phi = " << DLog::Prec{8} << lt << ", " << 90.0 - lp << " degrees." << '\n';
            log << '\t' << '\t' << '\t' << "Travel Time = " << travel_time << " seconds." << '\n';
            log << '\t' << '\t' << '\t' << "Celerity = " << e.v[GEOAC_EIG_CELERITY] << " km/s." << '\n';
            log << '\t' << '\t' << '\t' << "Amplitude = " << e.v[GEOAC_EIG_AMP_DB] << " dB." << '\n';
            log << '\t' << '\t' << '\t' << "Atmospheric Attenuation = " << -attenuation << " dB." << '\n';
            log << '\t' << '\t' << '\t' << "Arrival inclination = " << arrival_incl << " degrees." << '\n';
            log << '\t' << '\t' << '\t' << "Bearing to source = " << bearing_back << " degrees." << '\n';
            log << '\t' << '\t' << '\t' << "Back azimuth of arrival = " << back_az << " degrees." << '\n';
            log << '\t' << '\t' << '\t' << "Azimuth Deviation = " << back_az_dev << " degrees." << '\n' << '\n';
        } else {
            log << '\t' << "Eigenray identified:" << '\t' << "theta, phi = " << DLog::Prec{8} << lt << ", " << 90.0 - lp << " degrees." << '\n';
        }
        out.found.push_back(e);
        log.incr();
    }
};

// ---- 3-D Cartesian sets: GeoAc.Eigenray.cpp ----
double modify_d_theta_cart(double dr, double dr_dtheta){       // GeoAc.Eigenray.cpp:24-28
    double width = 2.0 * pow(dr_dtheta, 2);
    return d_theta_big - (d_theta_big - d_theta_small) * exp(-dr * dr / width);
}

struct CartesianFamily : Family {
    bool strat = true;                     // GeoAc_AtmoStrat: GeoAc3D (12-component rows) vs GeoAc3D.RngDep (18)
    double M_Comps[3] = {0, 0, 0};         // wind Mach numbers at the source (stratified set only, :130-134)
    const double* src() const { return site.src; }
    const double* rcv() const { return site.rcv; }
    void header(DLog& log, int n_bnc) const override { log << "Searching for " << n_bnc << " bounce eigenray(s) between " << site.prm.theta_min << " and " << site.prm.theta_max << "." << '\n'; }   // GeoAc3D_main.cpp:531-543
    void footer(DLog& log) const override { log << '\t' << "Identified " << DLog::Count{} << " eigenray(s)." << '\n'; }
    // GeoAc_EstimateEigenray: GeoAc.Eigenray.cpp:30-121
    double range_to_receiver() const override { return sqrt(pow(rcv()[0] - src()[0], 2) + pow(rcv()[1] - src()[1], 2)); }
    double first_azimuth() const override { return 180.0 / 3.14159 * atan2(rcv()[1] - src()[1], rcv()[0] - src()[0]); }
    double azimuth_from_north(double phi) const override { return 90.0 - phi; }
    double azimuth_estimate(double phi) const override { return phi; }
    bool scan_goes_on(double theta, double theta_max) const override { return theta <= theta_max; }
    void announce(DLog& log, double range0, double phi, double theta_min, double theta_max) const override {
        log << '\t' << "Estimating eigenray angles for source-receiver separated by " << range0 << " km, and azimuth " << 90.0 - phi;
        log << " degrees from N.  Inclination limits: [" << theta_min << ", " << theta_max << "]." << '\n';
    }
    double arrival(DLog& log, double theta, int bounces, bool left, const double* Rk, double range0) const override {
        const double xk = Rk[GEOAC_REC_STATE + 0], yk = Rk[GEOAC_REC_STATE + 1];
        if(site.verbose){
            log << '\t' << '\t' << "Ray launched at " << theta << " degrees arrives at range " << sqrt(pow(xk - src()[0], 2) + pow(yk - src()[1], 2));
            log << " km after " << bounces << " reflections." << '\t' << "Exact arrival at " << xk << " km East, " << yk << " km North" << '\n';
        }
        return left ? range0 : sqrt(pow(xk - src()[0], 2) + pow(yk - src()[1], 2));
    }
    double azimuth_deviation(const double* Rk) const override {
        const double xk = Rk[GEOAC_REC_STATE + 0], yk = Rk[GEOAC_REC_STATE + 1];
        return (atan2(rcv()[1] - src()[1], rcv()[0] - src()[0]) - atan2(yk - src()[1], xk - src()[0])) * 180.0 / Pi;
    }
    void verdict(DLog& log, bool ok, double d_phi, double limit) const override {
        if(ok) log << '\t' << '\t' << "Azimuth deviation = " << d_phi << ".  Less than " << limit << " degrees.  Estimates acceptable." << '\n' << '\n';
        else   log << '\t' << '\t' << "Azimuth deviation = " << d_phi << ".  Greater than " << limit << " degrees.  Compensating and searching inclinations again." << '\n' << '\n';
    }
    double next_step(double dr, double dr_dtheta) const override { return modify_d_theta_cart(dr, dr_dtheta); }
    void gave_up(DLog& log) const override { log << '\t' << '\t' << "Reached maximum inclination angle or iteration limit." << '\n'; }
    // GeoAc_3DEigenray_LM: GeoAc.Eigenray.cpp:123-335
    void launch_direction(Newton& N) const override {
        if(!strat) return;
        const double GeoAc_theta = N.lt * Pi / 180.0, GeoAc_phi = N.lp * Pi / 180.0;
        double nu0[3], M;
        nu0[0] = cos(GeoAc_theta) * cos(GeoAc_phi);
        nu0[1] = cos(GeoAc_theta) * sin(GeoAc_phi);
        nu0[2] = sin(GeoAc_theta);
        M = 1.0 + (nu0[0] * M_Comps[0] + nu0[1] * M_Comps[1] + nu0[2] * M_Comps[2]);
        N.nu0_xy[0] = nu0[0] / M;
        N.nu0_xy[1] = nu0[1] / M;
    }
    void left_region(DLog& log) const override { log << '\t' << "Ray path left propagation region." << '\n'; }
    void maxed_out(DLog& log) const override { log << '\t' << '\t' << '\t' << "Search for exact eigenray maxed out iterations.  No eigneray idenfied." << '\n' << '\n'; }
    double miss(DLog* log, Newton& N, const double* S, int bnc_cnt) const override {
        long double dx, dy;
        N.p = S[0]; dx = rcv()[0] - N.p;
        N.q = S[1]; dy = rcv()[1] - N.q;
        const double dr = (double)sqrtl(dx * dx + dy * dy);
        if(site.verbose && log) *log << '\t' << '\t' << "Arrival after " << bnc_cnt << " reflections at (" << (double)N.p << ", " << (double)N.q << "), distance to receiver = " << dr << " km." << '\n';
        return dr;
    }
    void full_step(Newton& N, const double* S) const override {
        const double theta_lim_step = 0.2, phi_lim_step = 0.2;
        long double dx = rcv()[0] - N.p, dy = rcv()[1] - N.q;
        long double dx_dt, dy_dt, dx_dp, dy_dp, det;
        if(strat){
            dx_dt = S[4] - N.nu0_xy[0] / S[3] * S[6];
            dy_dt = S[5] - N.nu0_xy[1] / S[3] * S[6];
            dx_dp = S[8] - N.nu0_xy[0] / S[3] * S[10];
            dy_dp = S[9] - N.nu0_xy[1] / S[3] * S[10];
        } else {
            dx_dt = S[6] - S[3] / S[5] * S[8];
            dy_dt = S[7] - S[4] / S[5] * S[8];
            dx_dp = S[12] - S[3] / S[5] * S[14];
            dy_dp = S[13] - S[4] / S[5] * S[14];
        }
        det = dx_dt * dy_dp - dx_dp * dy_dt;
        N.dlt = 1.0 / det * (dy_dp * dx - dx_dp * dy) * 180.0 / Pi;
        N.dlp = 1.0 / det * (dx_dt * dy - dy_dt * dx) * 180.0 / Pi;
        if(N.dlt > theta_lim_step)  N.dlt =  theta_lim_step;
        if(N.dlp > phi_lim_step)    N.dlp =  phi_lim_step;
        if(N.dlt < -theta_lim_step) N.dlt = -theta_lim_step;
        if(N.dlp < -phi_lim_step)   N.dlp = -phi_lim_step;
    }
    void identified(Segment& out, const Newton& N, const Request& fin, int bnc_cnt) const override {
        DLog& log = out.log;
        const double theta = N.lt, phi = N.lp;
        const double GeoAc_phi = phi * Pi / 180.0;
        const double* R = &fin.rec[((size_t)bnc_cnt) * GEOAC_REC_STRIDE];
        const double* Sk = R + GEOAC_REC_STATE;
        Eigenray e; memset(e.v, 0, sizeof e.v);
        const double travel_time = R[GEOAC_REC_TTIME], attenuation = R[GEOAC_REC_ATTEN];
        double back_az = strat ? (90.0 - GeoAc_phi * 180.0 / Pi) + 180.0 : 90.0 - atan2(-Sk[4], -Sk[3]) * 180.0 / Pi;
        double arrival_incl = R[GEOAC_REC_INCL];          // -asin(c(x_k, y_k, z_grnd) / c(src) nu_z) in both mains' records
        double az_to_src = 90.0 - atan2(src()[1] - rcv()[1], src()[0] - rcv()[0]) * 180.0 / Pi;
        double back_az_dev = back_az - az_to_src;
        while(back_az > 180.0)      back_az -= 360.0;
        while(back_az < -180.0)     back_az += 360.0;
        while(back_az_dev > 180.0)  back_az_dev -= 360.0;
        while(back_az_dev < -180.0) back_az_dev += 360.0;
        e.v[GEOAC_EIG_RCVR] = site.rcvr_index; e.v[GEOAC_EIG_INDEX] = 0;  /* numbered when the receiver's segments are put in order */ e.v[GEOAC_EIG_BOUNCES] = bnc_cnt;
        e.v[GEOAC_EIG_THETA] = theta; e.v[GEOAC_EIG_PHI] = 90.0 - phi;
        e.v[GEOAC_EIG_TTIME] = travel_time;
        e.v[GEOAC_EIG_CELERITY] = sqrt(pow(Sk[0] - src()[0], 2) + pow(Sk[1] - src()[1], 2)) / travel_time;
        e.v[GEOAC_EIG_AMP_DB] = 20.0 * log10(R[GEOAC_REC_AMP]);
        e.v[GEOAC_EIG_ATTEN_DB] = -attenuation;
        e.v[GEOAC_EIG_INCL] = arrival_incl; e.v[GEOAC_EIG_BEARING] = az_to_src; e.v[GEOAC_EIG_BACKAZ] = back_az; e.v[GEOAC_EIG_AZDEV] = back_az_dev;
        e.smp = fin.smp;
        e.v[GEOAC_EIG_NSMP] = (double)(e.smp.size() / GEOAC_SMP_STRIDE);
        if(!site.verbose) log << '\t' << "Eigenray identified:" << '\t' << "theta, phi = " << DLog::Prec{8} << theta << ", " << 90.0 - phi << " degrees." << '\n';
        if(site.verbose){
            log << '\t' << '\t' << "Eigenray Identified:" << '\n';
            log << '\t' << '\t' << '\t' << "theta, phi = " << DLog::Prec{8} << theta << ", " << 90.0 - phi << " degrees." << '\n';
            log << '\t' << '\t' << '\t' << "Travel Time = " << travel_time << " seconds." << '\n';
            log << '\t' << '\t' << '\t' << "Celerity = " << e.v[GEOAC_EIG_CELERITY] << " km/s." << '\n';
            log << '\t' << '\t' << '\t' << "Amplitude (geometric) = " << e.v[GEOAC_EIG_AMP_DB] << " dB." << '\n';
            log << '\t' << '\t' << '\t' << "Atmospheric Attenuation = " << -attenuation << " dB." << '\n';
            log << '\t' << '\t' << '\t' << "Arrival inclination = " << arrival_incl << " degrees." << '\n';
            log << '\t' << '\t' << '\t' << "Azimuth to source = " << az_to_src << '\n';
            log << '\t' << '\t' << '\t' << "Back Azimuth of arrival = " << back_az << '\n';
            log << '\t' << '\t' << '\t' << "Azimuth Deviation = " << back_az_dev << " degrees." << '\n' << '\n';
        }
        out.found.push_back(e);
        log.incr();
    }
};

// ================= tasks =================
// advance(): run until the task needs a ray (returns the request, which it owns and reads at its next call) or has ended (nullptr).
struct Task {
    virtual ~Task(){}
    virtual Request* advance(std::vector<std::unique_ptr<Task>>& spawned) = 0;
    virtual Request* also(){ return nullptr; }      // a second request for the same round (asked for right after advance() returned one)
};

// ---- GeoAc_EstimateEigenray as a state machine: up to five scans of the inclination range, the azimuth corrected between them ----
struct Estimator {
    enum State { PASS_BEGIN, SCAN_POSTED, RAY_NEXT, RAY_POSTED, PASS_END, DONE };
    const Family& fam; DLog& log;
    State st = PASS_BEGIN;
    const int bounces;
    double theta_min; const double theta_max;
    double range0, phi;
    int iterations = 0;
    double r = 0, r_prev = 0, d_theta = d_theta_big, d_phi = 10.0;
    bool theta_max_reached = false;
    double theta = 0;                       // scan position (adaptive passes)
    Request rq;                             // the posted scan or single ray
    // results
    bool ok = false;
    double theta_estimate, phi_estimate = 0, theta_next;

    Estimator(const Family& f, DLog& l, double tmin, double tmax, double tnext, int bnc)
        : fam(f), log(l), bounces(bnc), theta_min(tmin), theta_max(tmax), theta_estimate(tmax), theta_next(tnext) {
        range0 = fam.range_to_receiver();
        phi = fam.first_azimuth();
        if(fam.site.verbose) fam.announce(log, range0, phi, theta_min, theta_max);
    }
    void post(double th){ rq.th.push_back(th); rq.ph.push_back(fam.azimuth_from_north(phi)); }
    void fresh_request(){ rq = Request(); rq.bounces = bounces; rq.calc_amp = 0; rq.mode = 0; }

    enum Visit { GO_ON, ACCEPTED, RESCAN };
    // one ray of a scan, in the scan's order: does the arrival range cross the receiver's between this ray and the one before?
    Visit visit(double th, int idx){
        const double limit = fam.site.prm.azimuth_err_lim;
        const bool left = broke(rq, idx);
        const double* Rk = last_row(rq, idx);
        r = fam.arrival(log, th, bounces, left, Rk, range0);
        if(left) r_prev = range0;
        if((r - range0) * (r_prev - range0) < 0.0){
            if(iterations == 0) theta_next = th;
            d_phi = fam.azimuth_deviation(Rk);
            while(d_phi > 180.0){ d_phi -= 360.0; }
            while(d_phi < -180.0){ d_phi += 360.0; }
            if(fabs(d_phi) < limit){
                if(fam.site.verbose) fam.verdict(log, true, d_phi, limit);
                theta_estimate = th - d_theta;
                phi_estimate = fam.azimuth_estimate(phi);
                return ACCEPTED;
            }
            if(fam.site.verbose) fam.verdict(log, false, d_phi, limit);
            phi += d_phi * 0.9;
            theta_min = std::max(th - 7.5, theta_min);
            return RESCAN;
        }
        if(iterations >= 3){ d_theta = fam.next_step(r - range0, (r - r_prev) / (2.0 * d_theta)); }
        r_prev = r;
        return GO_ON;
    }

    // nullptr: finished (ok, theta_estimate, phi_estimate, theta_next are set)
    Request* advance(){
        const double limit = fam.site.prm.azimuth_err_lim;
        for(;;) switch(st){
        case PASS_BEGIN:
            if(!(fabs(d_phi) > limit && iterations < 5)){ if(fam.site.verbose) fam.gave_up(log); ok = false; st = DONE; return nullptr; }
            r = range0; r_prev = range0;
            fresh_request();
            if(iterations < 3){
                // the scan of this pass: with a fixed step all its rays are known now -> one request, replayed when it comes back
                for(double th = theta_min; fam.scan_goes_on(th, theta_max); th += d_theta) post(th);
                st = SCAN_POSTED;
                if(!rq.th.empty()) return &rq;
                break;
            }
            theta = theta_min;
            st = RAY_NEXT;
            break;
        case SCAN_POSTED: {
            Visit v = GO_ON;
            int j = 0;
            for(double th = theta_min; fam.scan_goes_on(th, theta_max); th += d_theta, j++){
                if(th + d_theta >= theta_max) theta_max_reached = true;
                v = visit(th, j);
                if(v != GO_ON) break;
            }
            if(v == ACCEPTED){ ok = true; st = DONE; return nullptr; }
            st = PASS_END;
            break;
        }
        case RAY_NEXT:                                              // the step depends on the previous arrival (:122): one ray at a time
            if(!fam.scan_goes_on(theta, theta_max)){ st = PASS_END; break; }
            if(theta + d_theta >= theta_max) theta_max_reached = true;
            fresh_request();
            post(theta);
            st = RAY_POSTED;
            return &rq;
        case RAY_POSTED: {
            const Visit v = visit(theta, 0);
            if(v == ACCEPTED){ ok = true; st = DONE; return nullptr; }
            if(v == RESCAN){ st = PASS_END; break; }
            theta += d_theta;
            st = RAY_NEXT;
            break;
        }
        case PASS_END:
            if(theta_max_reached){
                theta_next = theta_max;
                if(fam.site.verbose) fam.gave_up(log);
                ok = false; st = DONE; return nullptr;
            }
            iterations++;
            if(iterations >= 1 && iterations < 3){ d_theta = d_theta_big / 2.0; }
            st = PASS_BEGIN;
            break;
        case DONE:
            return nullptr;
        }
    }
};

// ---- GeoAc_3DEigenray_LM as a state machine: damped Newton steps on the launch angles until the arrival is within 100 m of the receiver.
//      A step that takes the arrival further away is undone and halved: the routine goes back to the point before (the same angles, bit for bit,
//      almost always) and tries again with 0.625 of the step - two rays, one after the other, for every such step, and the searches that end at
//      the iteration limit consist of little else.  So (a) every ray of the refinement is remembered and a ray whose angles it has seen is not
//      integrated again, and (b) beside the trial ray the request carries the trials that would follow if it (and the next, and the next) were
//      undone - computed by running the same transition on a copy of the state.  Consumed in the routine's order, with its log; a guess that
//      is not needed is a few lanes of a fan that was going to be launched anyway. ----
struct Refiner : Task {
    enum State { ITERATE, RAY_POSTED, FINAL_POSTED, DONE };
    enum { SPECULATE = 3 };                 // trials posted ahead
    const Family& fam; Segment& out;
    State st = ITERATE;
    const int bounces, iterate_limit;
    int n = 0;
    Newton N;
    Request rq;
    struct Known { double lt, lp; std::vector<double> rec; };      // a ray of this refinement (CalcAmp on, no samples) and its leg records
    std::vector<Known> known;
    std::vector<double> rq_lp;              // the azimuths of the posted rays as the routine carries them (the request has 90 - lp)
    Request fin;                            // the eigenray's own ray (with raypath samples), posted ahead beside a trial that is likely to be the last
    double fin_lt = 0, fin_lp = 0;
    bool fin_posted = false, fin_valid = false;

    Refiner(const Family& f, Segment& o, double lt, double lp, int bnc, int limit) : fam(f), out(o), bounces(bnc), iterate_limit(limit) {
        N.lt = lt; N.lp = lp;
        if(fam.site.verbose) out.log << '\t' << '\t' << "Searching for exact eigenray using auxiliary parameters." << '\n';
    }
    const double* lookup(double lt, double lp) const {
        for(const Known& k : known) if(k.lt == lt && k.lp == lp) return k.rec.data();
        return nullptr;
    }
    bool left_region(const double* rec) const { return rec[(size_t)bounces * GEOAC_REC_STRIDE + GEOAC_REC_VALID] == 0.0; }
    const double* last_state(const double* rec) const { return rec + (size_t)bounces * GEOAC_REC_STRIDE + GEOAC_REC_STATE; }   // solution[k][*] of the last leg

    // the routine's reaction to the arrival of the ray at (M.lt, M.lp), iteration m; log: the real run (nullptr: a look ahead)
    enum Next { TRIAL, FINAL, STOP };
    void undo_step(Newton& M) const { M.lt -= M.dlt * M.step_scalar; M.lp -= M.dlp * M.step_scalar; M.step_scalar /= 2.0; }
    bool step_too_small(const Newton& M) const { return sqrt(M.dlt * M.dlt + M.dlp * M.dlp) * M.step_scalar < 1.0e-12; }
    Next react(Newton& M, int& m, const double* rec, DLog* log) const {
        const bool verbose = fam.site.verbose && log;
        const double tolerance = 0.1;
        if(left_region(rec)){ if(verbose) fam.left_region(*log); return STOP; }
        const double* S = last_state(rec);
        M.dr = fam.miss(log, M, S, bounces);
        if(M.dr < tolerance) return FINAL;
        if(m > 0 && M.dr > M.dr_prev){
            undo_step(M);
            if(step_too_small(M)){
                if(verbose) *log << '\t' << '\t' << '\t' << "Step size too small, psuedo-critical ray path likely." << '\n' << '\n';
                return STOP;
            }
        } else {
            M.step_scalar = std::min(1.0, M.step_scalar * 1.25);
            fam.full_step(M, S);
            M.lt += M.dlt * M.step_scalar;
            M.lp += M.dlp * M.step_scalar;
            M.dr_prev = M.dr;
        }
        m++;
        return TRIAL;
    }
    void add_ray(double lt, double lp){ rq.th.push_back(lt); rq.ph.push_back(90.0 - lp); rq_lp.push_back(lp); }
    // the trials that follow if the ray just posted - iteration n, angles (N.lt, N.lp) - is undone, and the one after it, ...
    void look_ahead(){
        Newton M = N; int m = n;
        for(int k = 0; k < SPECULATE; k++){
            if(m == 0) return;                                       // the first ray is never undone
            undo_step(M);
            if(step_too_small(M)) return;
            m++;
            if(m == iterate_limit) return;
            fam.launch_direction(M);
            const double* back = lookup(M.lt, M.lp);                 // the point before, if the subtraction gave its angles back
            if(!back) return;
            if(react(M, m, back, nullptr) != TRIAL || m == iterate_limit) return;
            if(M.dr_prev != M.dr) return;                            // (it was undone again at once: not the pattern looked for)
            fam.launch_direction(M);
            bool posted = false;
            for(size_t i = 0; i < rq.th.size(); i++) posted = posted || (rq.th[i] == M.lt && rq_lp[i] == M.lp);
            if(!lookup(M.lt, M.lp) && !posted) add_ray(M.lt, M.lp);
        }
    }
    // ... and the trials that follow if the step just taken - the full step, cut to the routine's limit of 0.2 degrees in BOTH angles - is taken again:
    // far from the receiver the routine walks towards it in exactly these steps, one ray and one round each
    void walk_ahead(){
        const long double lim = 0.2;
        if(n == 0 || N.step_scalar != 1.0 || (N.dlt != lim && N.dlt != -lim) || (N.dlp != lim && N.dlp != -lim)) return;
        Newton M = N; int m = n;
        for(int k = 0; k < SPECULATE; k++){
            M.step_scalar = std::min(1.0, M.step_scalar * 1.25);
            M.lt += M.dlt * M.step_scalar;
            M.lp += M.dlp * M.step_scalar;
            m++;
            if(m == iterate_limit) return;
            bool posted = false;
            for(size_t i = 0; i < rq.th.size(); i++) posted = posted || (rq.th[i] == M.lt && rq_lp[i] == M.lp);
            if(!lookup(M.lt, M.lp) && !posted) add_ray(M.lt, M.lp);
        }
    }
    Request* also() override { return (fin_posted && st == RAY_POSTED) ? &fin : nullptr; }
    Request* advance(std::vector<std::unique_ptr<Task>>&) override {
        DLog& log = out.log;
        const bool verbose = fam.site.verbose;
        for(;;) switch(st){
        case ITERATE: {
            if(n == iterate_limit){ if(verbose) fam.maxed_out(log); st = DONE; break; }
            fam.launch_direction(N);
            if(verbose) log << '\t' << '\t' << "Plotting ray path with theta = " << N.lt << ", phi = " << 90.0 - N.lp;
            st = RAY_POSTED;
            if(lookup(N.lt, N.lp)) break;                            // seen before: no ray
            rq = Request(); rq_lp.clear(); rq.bounces = bounces; rq.calc_amp = 1; rq.mode = 0;
            add_ray(N.lt, N.lp);
            look_ahead();
            walk_ahead();
            // the arrival before this one missed the receiver by less than 500 m and the iteration converges quadratically there: this trial is
            // probably within the 100 m that end the search - its raypath ray goes out in the same round (another group, beside this one)
            fin_posted = fin_valid = false;
            if(n > 0 && N.dr_prev < 0.5){
                fin = Request(); fin.bounces = bounces; fin.calc_amp = 1; fin.mode = GEOAC_MODE_WRITE_RAYS;
                fin.th.push_back(N.lt); fin.ph.push_back(90.0 - N.lp);
                fin_lt = N.lt; fin_lp = N.lp; fin_posted = true;
            }
            return &rq;
        }
        case RAY_POSTED: {
            if(fin_posted){ fin_valid = true; fin_posted = false; }   // (it came back with this round)
            if(!rq.th.empty()){                                      // what came back: remember all of it
                const size_t per = (size_t)(bounces + 1) * GEOAC_REC_STRIDE;
                for(size_t i = 0; i < rq.th.size(); i++)
                    known.push_back(Known{ rq.th[i], rq_lp[i], std::vector<double>(rq.rec.begin() + i * per, rq.rec.begin() + (i + 1) * per) });
                rq = Request(); rq_lp.clear();
            }
            const double* rec = lookup(N.lt, N.lp);
            const Next nx = react(N, n, rec, &log);
            if(nx == STOP){ st = DONE; break; }
            if(nx == FINAL){
                st = FINAL_POSTED;
                if(fin_valid && fin_lt == N.lt && fin_lp == N.lp){ rq = fin; break; }      // already here
                rq = Request(); rq_lp.clear(); rq.bounces = bounces; rq.calc_amp = 1; rq.mode = GEOAC_MODE_WRITE_RAYS;      // the same ray once more, with its raypath
                add_ray(N.lt, N.lp);
                return &rq;
            }
            st = ITERATE;
            break;
        }
        case FINAL_POSTED:
            fam.identified(out, N, rq, bounces);
            st = DONE;
            break;
        case DONE:
            return nullptr;
        }
    }
};

// ---- one bounce count of one receiver: the -eig_search loop body (GeoAcGlobal_main.cpp:566-580, GeoAc3D_main.cpp:531-543).  Every estimate
//      and every refinement writes into a segment of its own; the refinements run as tasks beside the scan, the segments keep the reference's
//      order: header, E1, R1, E2, R2, ... ----
struct ScanChain : Task {
    const Family& fam;
    const int n_bnc;
    std::deque<Segment> segs;                // (deque: addresses stay valid while segments are appended)
    std::unique_ptr<Estimator> est;
    double theta_start, theta_next;
    ScanChain(const Family& f, int bnc) : fam(f), n_bnc(bnc), theta_start(f.site.prm.theta_min), theta_next(f.site.prm.theta_max) {
        segs.emplace_back();
        fam.header(segs.back().log, n_bnc);
    }
    Request* advance(std::vector<std::unique_ptr<Task>>& spawned) override {
        const geoac_eig_params& prm = fam.site.prm;
        for(;;){
            if(!est){
                if(!(theta_start < prm.theta_max)) return nullptr;
                segs.emplace_back();
                est.reset(new Estimator(fam, segs.back().log, theta_start, prm.theta_max, theta_next, n_bnc));
            }
            if(Request* rq = est->advance()) return rq;
            theta_next = est->theta_next;
            if(est->ok){
                segs.emplace_back();
                spawned.emplace_back(new Refiner(fam, segs.back(), est->theta_estimate, est->phi_estimate, n_bnc, prm.iterations));
            }
            theta_start = theta_next;
            est.reset();
        }
    }
};

// one receiver: its family (site data), its scan chains or its -eig_direct refinement, and the output put back in order
struct Receiver {
    std::unique_ptr<Family> fam;
    std::vector<ScanChain*> chains;          // (owned by the scheduler's task list)
    Segment direct;                          // -eig_direct: the one refinement's output
    std::ostringstream log;                  // the reference's cout for this receiver (sticky precision and all), rendered at the end
    std::vector<Eigenray> found;
    int eigenray_count = 0;
    void collect(Segment& g){
        g.log.render(log, eigenray_count);
        for(Eigenray& e : g.found){ e.v[GEOAC_EIG_INDEX] = (double)found.size(); found.push_back(e); }
    }
};

}  // namespace

struct geoac_eig_result {
    std::vector<double> eig;               // count x GEOAC_EIG_STRIDE
    std::vector<double> smp;
    std::vector<std::string> logs;
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// one (bounces, calc_amp, mode) group of a decision round as ONE fan launch on context `ctx`; results handed back to the group's requests
static int serve_group(geoac_ctx* ctx, const geoac_params& base, const std::tuple<int, int, int>& key, std::vector<Request*>& grp, uint64_t st[8]){
    geoac_params p = base;
    p.bounces = std::get<0>(key); p.calc_amp = std::get<1>(key); p.mode = std::get<2>(key);
    int rc = geoac_set_params(ctx, &p);
    if(rc) return rc;
    std::vector<double> th, ph;
    for(Request* r : grp){ th.insert(th.end(), r->th.begin(), r->th.end()); ph.insert(ph.end(), r->ph.begin(), r->ph.end()); }
    const int n = (int)th.size();
    const int legs = p.bounces + 1;
    std::vector<double> rec((size_t)n * legs * GEOAC_REC_STRIDE);
    uint64_t steps = 0;
    const auto t0 = std::chrono::steady_clock::now();
    rc = geoac_fan_run(ctx, n, th.data(), ph.data(), rec.data(), &steps);
    if(rc) return rc;
    if(st[7]){                                                      // (trace: GEOAC_EIG_TRACE)
        double longest = 0.0;
        for(int i = 0; i < n; i++){ double sum = 0.0; for(int l = 0; l < legs; l++) sum += rec[((size_t)i * legs + l) * GEOAC_REC_STRIDE + GEOAC_REC_STEPS]; longest = std::max(longest, sum); }
        fprintf(stderr, "    [group bounces %d amp %d mode %d] %d rays, longest %.0f steps, %.1f ms\n", p.bounces, p.calc_amp, p.mode, n, longest,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    st[0] += 1; st[1] += (uint64_t)n; st[2] += steps;
    {   // the launch's longest ray (all legs)
        double longest = 0.0;
        for(int i = 0; i < n; i++){
            double sum = 0.0;
            for(int l = 0; l < legs; l++) sum += rec[((size_t)i * legs + l) * GEOAC_REC_STRIDE + GEOAC_REC_STEPS];
            longest = std::max(longest, sum);
        }
        st[4] = std::max<uint64_t>(st[4], (uint64_t)longest);       // (the caller adds the round's critical path: its groups run side by side)
        if(p.calc_amp) st[5] += 1;
    }
    std::vector<double> smp;
    if(p.mode){
        int64_t ns = 0;
        rc = geoac_fan_sample_count(ctx, &ns); if(rc) return rc;
        smp.resize((size_t)std::max<int64_t>(ns, 1) * GEOAC_SMP_STRIDE);
        if(ns > 0){ rc = geoac_fan_fetch_samples(ctx, smp.data(), ns); if(rc) return rc; }
        smp.resize((size_t)ns * GEOAC_SMP_STRIDE);
    }
    size_t off = 0, sp = 0;
    const size_t nsmp = smp.size() / GEOAC_SMP_STRIDE;
    for(Request* r : grp){
        const size_t m = r->th.size();
        r->rec.assign(rec.begin() + off * legs * GEOAC_REC_STRIDE, rec.begin() + (off + m) * legs * GEOAC_REC_STRIDE);
        r->smp.clear();
        while(sp < nsmp && (size_t)smp[sp * GEOAC_SMP_STRIDE + GEOAC_SMP_RAY] < off + m){       // samples are sorted by ray
            size_t b = r->smp.size();
            r->smp.insert(r->smp.end(), smp.begin() + sp * GEOAC_SMP_STRIDE, smp.begin() + (sp + 1) * GEOAC_SMP_STRIDE);
            r->smp[b + GEOAC_SMP_RAY] -= (double)off;
            sp++;
        }
        off += m;
    }
    return 0;
}

// Integrate all pending requests of a decision round: one fan launch per (bounces, calc_amp, mode) group, the groups SIDE BY SIDE - each on a
// context of its own (the caller's, and clones of it that share its atmosphere tables: geoac_clone), from a host thread of its own.  A
// group of a search round is a handful of rays - a few waves on a chip of 1 024 wave slots - and lasts as long as its longest ray, so the
// groups of a round overlap almost completely: the round costs its longest group, not the sum of them.
static int serve(std::vector<geoac_ctx*>& ctxs, const geoac_params& base, std::vector<Request*>& reqs, geoac_eig_result* res, bool trace_groups = false, bool merge = true){
    // Requests of one round that launch the SAME rays and differ only in the bounce count are integrated once, with the largest count: a ray with b
    // bounces goes through the states of the ray with fewer, its records of legs 0 .. a ARE the a-bounce ray's (bit for bit:
    // test_leg_records_do_not_depend_on_the_number_of_bounces).  The first inclination scans of a search are such requests - one per bounce count
    // over the same inclinations towards the same receiver (GeoAc.Eigenray.Global.cpp:46-70 runs them one after the other): config 5's first round
    // is one fan instead of three.  Arrivals only (mode 0: sample rows are per launch).
    std::vector<std::pair<Request*, Request*>> riders;              // (request, the request whose launch carries its rays)
    {
        std::vector<Request*> lead;                                // per distinct (calc_amp, ray list): the request with the most bounces so far
        std::vector<Request*> kept;
        for(Request* r : reqs){
            if(r->mode != 0 || !merge){ kept.push_back(r); continue; }
            Request** slot = nullptr;
            for(Request*& l : lead) if(l->calc_amp == r->calc_amp && l->th == r->th && l->ph == r->ph){ slot = &l; break; }
            if(!slot){ lead.push_back(r); continue; }
            if(r->bounces > (*slot)->bounces){ riders.emplace_back(*slot, nullptr); *slot = r; }
            else riders.emplace_back(r, nullptr);
        }
        for(auto& rd : riders)
            for(Request* l : lead) if(l->calc_amp == rd.first->calc_amp && l->th == rd.first->th && l->ph == rd.first->ph){ rd.second = l; break; }
        for(Request* l : lead) kept.push_back(l);
        reqs.swap(kept);                                           // (the riders are filled in below, after the launches)
    }
    std::map<std::tuple<int, int, int>, std::vector<Request*>> groups;
    for(Request* r : reqs) groups[std::make_tuple(r->bounces, r->calc_amp, r->mode)].push_back(r);
    while(ctxs.size() < groups.size() && ctxs.size() < 8){
        geoac_ctx* c = nullptr;
        int rc = geoac_clone(ctxs[0], &c);
        if(rc) break;                                              // (no clone: the groups take turns on the contexts there are)
        ctxs.push_back(c);
    }
    // (a HOME context per kind of group - so that a context's buffers, which only grow, and grow by hipFree + hipMalloc, fit its kind of fan - was
    //  measured: nine kinds for bounces 0 .. 2, nine contexts to create and warm instead of five or six: config 5 1.75 -> 2.13 s.  Groups go to the
    //  first free context.)
    std::vector<std::pair<const std::tuple<int, int, int>*, std::vector<Request*>*>> work;
    for(auto& g : groups) work.emplace_back(&g.first, &g.second);
    std::atomic<size_t> next{0};
    std::atomic<int> first_rc{0};
    std::mutex mu;
    uint64_t round_crit = 0;
    auto worker = [&](geoac_ctx* c){
        uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, trace_groups ? 1ull : 0ull};
        for(;;){
            const size_t i = next.fetch_add(1);
            if(i >= work.size() || first_rc.load()) break;
            int rc = serve_group(c, base, *work[i].first, *work[i].second, st);
            if(rc){ int z = 0; first_rc.compare_exchange_strong(z, rc); break; }
        }
        std::lock_guard<std::mutex> lk(mu);
        res->stats[0] += st[0]; res->stats[1] += st[1]; res->stats[2] += st[2]; res->stats[5] += st[5];
        round_crit = std::max(round_crit, st[4]);
    };
    const size_t nw = std::min(ctxs.size(), work.size());
    std::vector<std::thread> th;
    for(size_t w = 1; w < nw; w++) th.emplace_back(worker, ctxs[w]);
    worker(ctxs[0]);
    for(auto& t : th) t.join();
    res->stats[4] += round_crit;
    if(!first_rc.load()){
        for(auto& rd : riders){                                    // legs 0 .. a of the carrier's records
            Request* r = rd.first; const Request* c = rd.second;
            const size_t n = r->th.size(), la = (size_t)r->bounces + 1, lc = (size_t)c->bounces + 1;
            r->rec.resize(n * la * GEOAC_REC_STRIDE);
            for(size_t i = 0; i < n; i++)
                std::copy(c->rec.begin() + (i * lc) * GEOAC_REC_STRIDE, c->rec.begin() + (i * lc + la) * GEOAC_REC_STRIDE, r->rec.begin() + (i * la) * GEOAC_REC_STRIDE);
            r->smp.clear();
        }
    }
    return first_rc.load();
}

static int run_all(geoac_ctx* ctx, const geoac_eig_params* ep, int n_rcvr, const double* rcvr, bool direct,
                   const double* theta_est, const double* phi_est, int bounces, geoac_eig_result** out){
    if(!ctx || !ep || n_rcvr < 1 || !rcvr || !out) return GEOAC_E_INVALID;
    geoac_params base;
    int rc = geoac_get_params(ctx, &base);
    if(rc) return rc;
    int eqset = 0;
    rc = geoac_get_eqset(ctx, &eqset);
    if(rc) return rc;
    const bool sph = (eqset == GEOAC_EQ_GLOBAL || eqset == GEOAC_EQ_GLOBAL_RNGDEP);
    const bool cart = (eqset == GEOAC_EQ_3D || eqset == GEOAC_EQ_3D_RNGDEP);
    if(!sph && !cart) return GEOAC_E_UNSUPPORTED;
    double mach[3] = {0, 0, 0};
    if(eqset == GEOAC_EQ_3D){                                       // u/c, v/c, w/c at the source (GeoAc.Eigenray.cpp:130-134)
        double m4[4];
        rc = geoac_medium_1d(ctx, std::max(base.src[2], base.z_grnd), m4);
        if(rc) return rc;
        mach[0] = m4[1] / m4[0]; mach[1] = m4[2] / m4[0]; mach[2] = 0.0 / m4[0];
    }
    // ---- the receivers and their first tasks ----
    std::vector<Receiver> R((size_t)n_rcvr);
    std::vector<std::unique_ptr<Task>> all;                         // every task ever made (the scan chains hold the output until the end)
    std::vector<Task*> active;
    for(int i = 0; i < n_rcvr; i++){
        Receiver& rv = R[(size_t)i];
        if(sph){
            SphericalFamily* q = new SphericalFamily(); q->geo.r_earth = base.r_earth; rv.fam.reset(q);
            q->site.src[0] = base.src[1]; q->site.src[1] = base.src[2]; q->site.src[2] = std::max(base.src[0], base.z_grnd);   // Source_Loc = (lat, lon, max(z, z_grnd))
        } else {
            CartesianFamily* q = new CartesianFamily(); q->strat = (eqset == GEOAC_EQ_3D); rv.fam.reset(q);
            for(int c = 0; c < 3; c++) q->M_Comps[c] = mach[c];
            q->site.src[0] = base.src[0]; q->site.src[1] = base.src[1]; q->site.src[2] = std::max(base.src[2], base.z_grnd);   // Source_Loc = (x, y, max(z, z_grnd))
        }
        Site& s = rv.fam->site;
        s.eqset = eqset; s.rcvr_index = i; s.z_grnd = base.z_grnd;
        s.rcv[0] = rcvr[2 * i]; s.rcv[1] = rcvr[2 * i + 1];
        s.prm = *ep; s.verbose = ep->verbose != 0;
        if(direct){
            all.emplace_back(new Refiner(*rv.fam, rv.direct, theta_est[i], 90.0 - phi_est[i], bounces, ep->iterations));
            active.push_back(all.back().get());
        } else {
            for(int b = ep->bnc_min; b <= ep->bnc_max; b++){
                ScanChain* c = new ScanChain(*rv.fam, b);
                all.emplace_back(c); active.push_back(c); rv.chains.push_back(c);
            }
        }
    }
    // ---- rounds: every task runs until it needs a ray; the rays of the round are integrated together; the tasks read their records ----
    geoac_eig_result* res = new geoac_eig_result();
    std::vector<geoac_ctx*> ctxs{ ctx };                          // the caller's context and, made on demand, clones of it (serve)
    int err = 0;
    const char* dbg = getenv("GEOAC_DEBUG_ENV");
    const bool trace = dbg && dbg[0] == '1' && getenv("GEOAC_EIG_TRACE") != nullptr;
    const char* mg = (dbg && dbg[0] == '1') ? getenv("GEOAC_EIG_MERGE") : nullptr;      // (A/B: 0 = every request its own rays)
    const bool merge = !(mg && mg[0] == '0');
    while(!active.empty() && !err){
        std::vector<Request*> batch;
        std::vector<Task*> waiting;
        for(size_t i = 0; i < active.size(); i++){                  // (tasks spawned in this round join it)
            std::vector<std::unique_ptr<Task>> spawned;
            Request* rq = active[i]->advance(spawned);
            for(auto& t : spawned){ active.push_back(t.get()); all.push_back(std::move(t)); }
            if(rq){ batch.push_back(rq); waiting.push_back(active[i]); if(Request* more = active[i]->also()) batch.push_back(more); }
        }
        if(batch.empty()) break;
        if(trace){                                                  // (GEOAC_DEBUG_ENV=1 GEOAC_EIG_TRACE=1) what this round is made of
            size_t scans = 0, scan_rays = 0, singles = 0, newton = 0, finals = 0;
            for(Request* r : batch){
                if(r->mode) finals++; else if(r->calc_amp) newton++; else if(r->th.size() > 1){ scans++; scan_rays += r->th.size(); } else singles++;
            }
            fprintf(stderr, "[eig round %llu] tasks %zu: %zu scans (%zu rays), %zu single scan rays, %zu refinement rays, %zu eigenray rays\n",
                    (unsigned long long)res->stats[3], active.size(), scans, scan_rays, singles, newton, finals);
        }
        const auto t0 = std::chrono::steady_clock::now();
        err = serve(ctxs, base, batch, res, trace, merge);
        if(trace) fprintf(stderr, "[eig round %llu] integrated in %.1f ms\n", (unsigned long long)res->stats[3], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        res->stats[3] += 1;
        active.swap(waiting);
    }
    for(size_t w = 1; w < ctxs.size(); w++) geoac_destroy(ctxs[w]);
    geoac_set_params(ctx, &base);                                  // restore the caller's bounces / calc_amp / mode
    if(err){ delete res; return err; }
    // ---- the output in the reference's order ----
    for(int i = 0; i < n_rcvr; i++){
        Receiver& rv = R[(size_t)i];
        if(direct) rv.collect(rv.direct);
        else {
            for(ScanChain* c : rv.chains) for(Segment& g : c->segs) rv.collect(g);
            Segment tail;
            rv.fam->footer(tail.log);
            tail.log.render(rv.log, rv.eigenray_count);
        }
        for(Eigenray& e : rv.found){
            e.v[GEOAC_EIG_SMP0] = (double)(res->smp.size() / GEOAC_SMP_STRIDE);
            const size_t idx = res->eig.size() / GEOAC_EIG_STRIDE;
            res->eig.insert(res->eig.end(), e.v, e.v + GEOAC_EIG_STRIDE);
            for(size_t q = 0; q + GEOAC_SMP_STRIDE <= e.smp.size(); q += GEOAC_SMP_STRIDE){
                size_t b = res->smp.size();
                res->smp.insert(res->smp.end(), e.smp.begin() + q, e.smp.begin() + q + GEOAC_SMP_STRIDE);
                res->smp[b + GEOAC_SMP_RAY] = (double)idx;
            }
        }
        res->logs.push_back(rv.log.str());
    }
    *out = res;
    return 0;
}

extern "C" {

int geoac_eig_default_params(geoac_eig_params* p){
    if(!p) return GEOAC_E_INVALID;
    p->theta_min = 0.5; p->theta_max = 45.0; p->bnc_min = 0; p->bnc_max = 0; p->iterations = 25; p->azimuth_err_lim = 2.0; p->verbose = 0;
    return 0;
}
int geoac_eig_search(geoac_ctx* ctx, const geoac_eig_params* p, int n_rcvr, const double* rcvr, geoac_eig_result** out){
    return run_all(ctx, p, n_rcvr, rcvr, false, nullptr, nullptr, 0, out);
}
int geoac_eig_direct(geoac_ctx* ctx, const geoac_eig_params* p, int n_rcvr, const double* rcvr,
                     const double* theta_est, const double* phi_est, int bounces, geoac_eig_result** out){
    if(!theta_est || !phi_est || bounces < 0) return GEOAC_E_INVALID;
    return run_all(ctx, p, n_rcvr, rcvr, true, theta_est, phi_est, bounces, out);
}
int64_t geoac_eig_count(const geoac_eig_result* r){ return r ? (int64_t)(r->eig.size() / GEOAC_EIG_STRIDE) : 0; }
int geoac_eig_fetch(const geoac_eig_result* r, double* eig){
    if(!r || !eig) return GEOAC_E_INVALID;
    memcpy(eig, r->eig.data(), r->eig.size() * sizeof(double));
    return 0;
}
int64_t geoac_eig_sample_count(const geoac_eig_result* r){ return r ? (int64_t)(r->smp.size() / GEOAC_SMP_STRIDE) : 0; }
int geoac_eig_fetch_samples(const geoac_eig_result* r, double* smp){
    if(!r || !smp) return GEOAC_E_INVALID;
    memcpy(smp, r->smp.data(), r->smp.size() * sizeof(double));
    return 0;
}
const char* geoac_eig_log(const geoac_eig_result* r, int rcvr){
    if(!r || rcvr < 0 || rcvr >= (int)r->logs.size()) return "";
    return r->logs[(size_t)rcvr].c_str();
}
int geoac_eig_stats(const geoac_eig_result* r, uint64_t stats[4]){
    if(!r || !stats) return GEOAC_E_INVALID;
    for(int i = 0; i < 4; i++) stats[i] = r->stats[i];
    return 0;
}
int geoac_eig_stats_ex(const geoac_eig_result* r, uint64_t stats[8]){
    if(!r || !stats) return GEOAC_E_INVALID;
    for(int i = 0; i < 8; i++) stats[i] = r->stats[i];
    return GEOAC_OK;
}
void geoac_eig_free(geoac_eig_result* r){ delete r; }

}  // extern "C"
