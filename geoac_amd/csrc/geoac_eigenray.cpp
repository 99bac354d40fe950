// geoac_eigenray.cpp - batched eigenray searches on top of the ray-fan C ABI (include/geoac_eig.h): spherical sets
// (GeoAc.Eigenray.Global.cpp) and 3-D Cartesian sets (GeoAc.Eigenray.cpp).
//
// Every receiver's search is the reference's straight-line logic (GeoAc.Eigenray.Global.cpp) running in its own host thread; wherever
// the reference would propagate a ray the thread posts a request and sleeps.  When every live search is waiting, the calling thread
// groups the requests by (bounces, CalcAmp, output mode), integrates each group as ONE fan launch on the GPU and wakes the searches
// with their arrival records.  An inclination scan whose step does not depend on earlier arrivals (the first three passes of
// GeoAc_EstimateEigenray) is requested as a whole and replayed in the reference's order afterwards.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
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
    bool done = false;
    int  error = 0;
};

struct Eigenray { double v[GEOAC_EIG_STRIDE]; std::vector<double> smp; };

struct Shared {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<Request*> pending;
    int active = 0;                        // searches still running
    int waiting = 0;                       // searches blocked in trace()
    std::atomic<bool> failed{false};     // written by the coordinator under the mutex, read by the scan-chain threads without it
};

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

// Deferred log: what a search task would have written to the reference's cout, kept as operations and rendered at the end
// into ONE stream per receiver in the reference's order - the tasks of a receiver (one scan chain per bounce count, one task per
// refinement) run concurrently, but a std::setprecision of an earlier block must still stick to the later ones, and the running
// eigenray number is only known once the earlier blocks are complete.
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

struct Eigenray;
struct Segment;                            // one task's output: its log and the eigenrays it identified
thread_local DLog* tl_log = nullptr;
thread_local std::vector<Eigenray>* tl_found = nullptr;
#define LOG (*tl_log)
#define FOUND (*tl_found)

struct SearchBase {
    Shared* sh = nullptr;
    int eqset = GEOAC_EQ_GLOBAL;
    int rcvr_index = 0;
    double src[3] = {0, 0, 0};             // Source_Loc of the reference: (lat, lon [deg], z) or (x, y, z) [km]
    double rcv[2] = {0, 0};
    double z_grnd = 0.0;
    geoac_eig_params prm{};
    bool verbose = false;
    std::ostringstream log;                // the reference's cout for this receiver (sticky precision and all), rendered at the end
    std::vector<Eigenray> found;
    int eigenray_count = 0;
    bool self_released = false;            // run_search took this thread out of Shared::active before waiting for its tasks
    virtual ~SearchBase(){}
    virtual void run_direct(double theta_est, double phi_from_north, int bounces) = 0;
    // the pieces of the reference's -eig_search loop (GeoAcGlobal_main.cpp:566-580, GeoAc3D_main.cpp:531-543) the two families differ in
    virtual void header(int n_bnc) = 0;
    virtual void footer() = 0;
    virtual bool estimate(double theta_min, double theta_max, double& theta_estimate, double& phi_estimate, double& theta_next, int bounces) = 0;
    virtual void refine(double& lt, double& lp, int bnc_cnt, int iterate_limit) = 0;

    // ---- tasks.  The reference runs, per bounce count, estimate -> refine -> estimate -> ... one ray at a time.  Nothing an
    //      estimate or a refinement computes depends on another bounce count, and a refinement depends only on the estimate it
    //      starts from; so every bounce count is its own scan chain and every refinement its own task, all posting rays to the same
    //      rounds (fewer, fuller fan launches), and the output is put back in the reference's order afterwards. ----
    struct Segment { DLog log; std::vector<Eigenray> found; };
    void task_started(int n = 1){ std::unique_lock<std::mutex> lk(sh->mu); sh->active += n; }
    void task_ended(){ std::unique_lock<std::mutex> lk(sh->mu); sh->active--; sh->cv_work.notify_all(); }
    static std::atomic<int>& live_tasks(){ static std::atomic<int> n{0}; return n; }

    void run_search(){
        const int nb = std::max(0, prm.bnc_max - prm.bnc_min + 1);
        std::vector<std::deque<Segment>> segs((size_t)nb);           // per bounce count, in the reference's order: header, E1, R1, E2, R2, ...
        std::vector<std::thread> chains;
        task_started(nb);
        for(int b = 0; b < nb; b++){
            chains.emplace_back([this, b, &segs]{
                const int n_bnc = prm.bnc_min + b;
                std::deque<Segment>& mine = segs[(size_t)b];
                std::vector<std::thread> refiners;
                mine.emplace_back();
                tl_log = &mine.back().log; tl_found = &mine.back().found;
                header(n_bnc);
                double theta_start = prm.theta_min, theta_next = prm.theta_max, theta_est = 0, phi_est = 0;
                while(theta_start < prm.theta_max && !sh->failed){
                    mine.emplace_back();
                    tl_log = &mine.back().log; tl_found = &mine.back().found;
                    bool ok = estimate(theta_start, prm.theta_max, theta_est, phi_est, theta_next, n_bnc);
                    if(sh->failed) break;
                    if(ok){
                        mine.emplace_back();
                        Segment* seg = &mine.back();                 // deque: the address stays valid while later segments are appended
                        const double lt0 = theta_est, lp0 = phi_est;
                        if(live_tasks().load() < 384){
                            live_tasks()++;
                            task_started();
                            refiners.emplace_back([this, seg, lt0, lp0, n_bnc]{
                                tl_log = &seg->log; tl_found = &seg->found;
                                double lt = lt0, lp = lp0;
                                refine(lt, lp, n_bnc, prm.iterations);
                                live_tasks()--;
                                task_ended();
                            });
                        } else {                                       // very many receivers: refine in line, as the reference does
                            tl_log = &seg->log; tl_found = &seg->found;
                            double lt = lt0, lp = lp0;
                            refine(lt, lp, n_bnc, prm.iterations);
                        }
                    }
                    theta_start = theta_next;
                }
                task_ended();                                          // nothing more to trace from this thread: it only waits now
                for(auto& t : refiners) t.join();
            });
        }
        task_ended();                                                  // this receiver's own thread only waits for its chains
        self_released = true;
        for(auto& t : chains) t.join();
        // ---- put the output in the reference's order ----
        for(auto& per_bnc : segs) for(Segment& g : per_bnc){
            g.log.render(log, eigenray_count);
            for(Eigenray& e : g.found){ e.v[GEOAC_EIG_INDEX] = (double)found.size(); found.push_back(e); }
        }
        Segment tail;
        tl_log = &tail.log; tl_found = &tail.found;
        footer();
        tail.log.render(log, eigenray_count);
    }
    // -eig_direct: one refinement, on this thread
    void direct_refine(double lt, double lp, int bounces){
        Segment g;
        tl_log = &g.log; tl_found = &g.found;
        refine(lt, lp, bounces, prm.iterations);
        g.log.render(log, eigenray_count);
        for(Eigenray& e : g.found){ e.v[GEOAC_EIG_INDEX] = (double)found.size(); found.push_back(e); }
    }

    // ---- post a request and wait for the coordinator ----
    bool trace(Request& rq){
        std::unique_lock<std::mutex> lk(sh->mu);
        rq.done = false;
        sh->pending.push_back(&rq);
        sh->waiting++;
        sh->cv_work.notify_all();
        sh->cv_done.wait(lk, [&]{ return rq.done || sh->failed; });
        return rq.done && rq.error == 0 && !sh->failed;
    }
    // outcome of ray i of a request: BreakCheck of the reference's leg loop, last row solution[k][*]
    static bool broke(const Request& rq, int i){
        const int legs = rq.bounces + 1;
        const double* R = &rq.rec[((size_t)i * legs + (legs - 1)) * GEOAC_REC_STRIDE];
        return R[GEOAC_REC_VALID] == 0.0;
    }
    static const double* last_row(const Request& rq, int i){
        const int legs = rq.bounces + 1;
        // the row the reference reads after a break is the breaking leg's last row; only used for messages there
        int l = legs - 1;
        while(l > 0 && rq.rec[((size_t)i * legs + l) * GEOAC_REC_STRIDE + GEOAC_REC_STEPS] == 0.0) l--;
        return &rq.rec[((size_t)i * legs + l) * GEOAC_REC_STRIDE];
    }
};

// ================= spherical sets: GeoAc.Eigenray.Global.cpp =================
struct Search : SearchBase {
    Geo geo{6370.0};

    // ---- GeoAc_EstimateEigenray: GeoAc.Eigenray.Global.cpp:46-136 ----
    bool estimate(double theta_min, double theta_max, double& theta_estimate, double& phi_estimate, double& theta_next, int bounces) override {
        const double azimuth_error_limit = prm.azimuth_err_lim;
        double GC_r_rcvr = geo.gc_distance(src[0], src[1], rcv[0], rcv[1]);
        double phi = geo.bearing(src[0], src[1], rcv[0], rcv[1]);
        if(verbose){
            LOG << '\t' << "Estimating eigenray angles for source-receiver separated by great circle distance " << GC_r_rcvr << " km, and azimuth " << phi;
            LOG << " degrees from N.  Inclination limits: [" << theta_min << ", " << theta_max << "]." << '\n';
        }
        int iterations = 0;
        theta_estimate = theta_max;
        double r, r_prev, d_theta = d_theta_big, d_phi = 10.0;
        bool theta_max_reached = false;
        while(fabs(d_phi) > azimuth_error_limit && iterations < 5){
            r = GC_r_rcvr; r_prev = GC_r_rcvr;
            // the scan of this pass: with a fixed step all its rays are known now -> one request, replayed below
            Request scan; scan.bounces = bounces; scan.calc_amp = 0; scan.mode = 0;
            const bool batched = (iterations < 3);
            if(batched){
                for(double theta = theta_min; theta < theta_max; theta += d_theta){ scan.th.push_back(theta); scan.ph.push_back(phi); }
                if(!scan.th.empty() && !trace(scan)) return false;
            }
            int j = 0;
            bool crossed = false;
            for(double theta = theta_min; theta < theta_max; theta += d_theta, j++){
                if(theta + d_theta >= theta_max) theta_max_reached = true;
                Request one; const Request* rq = &scan; int idx = j;
                if(!batched){                                           // step depends on the previous arrival (:122): one ray at a time
                    one.bounces = bounces; one.calc_amp = 0; one.mode = 0; one.th.push_back(theta); one.ph.push_back(phi);
                    if(!trace(one)) return false;
                    rq = &one; idx = 0;
                }
                const bool BreakCheck = broke(*rq, idx);
                const double* Rk = last_row(*rq, idx);
                const double lat_k = Rk[GEOAC_REC_STATE + 1] * 180.0 / Pi, lon_k = Rk[GEOAC_REC_STATE + 2] * 180.0 / Pi;
                if(BreakCheck){ r = GC_r_rcvr; r_prev = GC_r_rcvr; }
                else r = geo.gc_distance(src[0], src[1], lat_k, lon_k);
                if(verbose){
                    LOG << '\t' << '\t' << "Ray launched at inclination=" << (theta * Pi / 180.0) * 180.0 / Pi << " degrees arrives at range " << r;
                    LOG << " km after " << bounces << " bounces.  Exact arrival at " << lat_k << " degrees N latitude, " << lon_k << " degrees E longitude" << '\n';
                }
                if((r - GC_r_rcvr) * (r_prev - GC_r_rcvr) < 0.0){
                    if(iterations == 0) theta_next = theta;
                    d_phi  = geo.bearing(src[0], src[1], rcv[0], rcv[1]);
                    d_phi -= geo.bearing(src[0], src[1], lat_k, lon_k);
                    while(d_phi > 180.0){ d_phi -= 360.0; }
                    while(d_phi < -180.0){ d_phi += 360.0; }
                    if(fabs(d_phi) < azimuth_error_limit){
                        if(verbose) LOG << '\t' << '\t' << "Azimuth deviation less than " << azimuth_error_limit << " degrees.  Estimates acceptable." << '\n' << '\n';
                        theta_estimate = theta - d_theta;
                        phi_estimate = 90.0 - phi;
                        return true;
                    } else {
                        if(verbose) LOG << '\t' << '\t' << "Azimuth deviation greater than " << azimuth_error_limit << " degrees.  Compensating and searching inclinations again." << '\n' << '\n';
                        phi += d_phi * 0.9;
                        theta_min = std::max(theta - 7.5, theta_min);
                    }
                    crossed = true;
                    break;
                }
                if(iterations >= 3){ d_theta = modify_d_theta(r - GC_r_rcvr, (r - r_prev) / (2.0 * d_theta)); }
                r_prev = r;
            }
            (void)crossed;
            if(theta_max_reached){
                theta_next = theta_max;
                break;
            }
            iterations++;
            if(iterations >= 1 && iterations < 3){ d_theta = d_theta_big / 2.0; }
        }
        if(verbose) LOG << '\t' << '\t' << "Reached maximum inclination angle or iteration limit." << '\n' << '\n';
        return false;
    }

    // ---- GeoAc_3DEigenray_LM: GeoAc.Eigenray.Global.cpp:139-319 ----
    void refine(double& lt, double& lp, int bnc_cnt, int iterate_limit) override {
        double dr, dr_prev = 10000.0;
        const double tolerance = 0.1;
        const double lt_lim_step = 0.2, lp_lim_step = 0.2;
        double step_scalar = 1.0;
        long double lat, lon, d_lat, d_lon, d_lat_dlt, d_lon_dlt, d_lat_dlp, d_lon_dlp, det, dlt = 0, dlp = 0;
        if(verbose) LOG << '\t' << '\t' << "Searching for exact eigenray using auxiliary parameters." << '\n';
        for(int n = 0; n <= iterate_limit; n++){
            if(n == iterate_limit){
                if(verbose){ LOG << '\t' << '\t' << '\t' << "Search for exact eigenray maxed out iterations.  No eigneray idenfied." << '\n'; }
                break;
            }
            Request rq; rq.bounces = bnc_cnt; rq.calc_amp = 1; rq.mode = 0;
            rq.th.push_back(lt); rq.ph.push_back(90.0 - lp);
            if(verbose) LOG << '\t' << '\t' << "Plotting ray path with theta = " << lt << ", phi = " << 90.0 - lp;
            if(!trace(rq)) return;
            if(broke(rq, 0)) break;
            const double* S = &rq.rec[((size_t)bnc_cnt) * GEOAC_REC_STRIDE + GEOAC_REC_STATE];        // solution[k][*] of the last leg
            lat = S[1]; lon = S[2];
            dr = geo.gc_distance((double)(lat * 180.0 / Pi), (double)(lon * 180.0 / Pi), rcv[0], rcv[1]);
            if(verbose) LOG << '\t' << '\t' << "Arrival at (" << DLog::Prec{8} << (double)(lat * 180.0 / Pi) << ", " << (double)(lon * 180.0 / Pi) << "), distance to receiver = " << dr << " km." << '\n';

            if(dr < tolerance){
                // the reference re-propagates and accumulates travel time / attenuation with the raypath-writing loop (:198-238):
                // same ray again with WriteRays on (segment form of Q7, samples every 25th step)
                Request fin; fin.bounces = bnc_cnt; fin.calc_amp = 1; fin.mode = GEOAC_MODE_WRITE_RAYS;
                fin.th.push_back(lt); fin.ph.push_back(90.0 - lp);
                if(!trace(fin)) return;
                const double* R = &fin.rec[((size_t)bnc_cnt) * GEOAC_REC_STRIDE];
                const double* Sk = R + GEOAC_REC_STATE;
                Eigenray e; memset(e.v, 0, sizeof e.v);
                const double travel_time = R[GEOAC_REC_TTIME], attenuation = R[GEOAC_REC_ATTEN];
                // arrival inclination: -asin(c_k / c_src nu_r) (:241); the fan record of the range-dependent main carries the opposite sign (Q10)
                double arrival_incl = (eqset == GEOAC_EQ_GLOBAL_RNGDEP) ? -R[GEOAC_REC_INCL] : R[GEOAC_REC_INCL];
                double bearing_back = geo.bearing(rcv[0], rcv[1], src[0], src[1]);
                double back_az = 90.0 - atan2(-Sk[4], -Sk[5]) * 180.0 / Pi;
                double back_az_dev = back_az - bearing_back;
                if(back_az_dev > 180.0)  back_az_dev -= 360.0;
                if(back_az_dev < -180.0) back_az_dev += 360.0;
                e.v[GEOAC_EIG_RCVR] = rcvr_index; e.v[GEOAC_EIG_INDEX] = 0;  /* numbered when the receiver's segments are put in order */ e.v[GEOAC_EIG_BOUNCES] = bnc_cnt;
                e.v[GEOAC_EIG_THETA] = lt; e.v[GEOAC_EIG_PHI] = 90.0 - lp;
                e.v[GEOAC_EIG_TTIME] = travel_time;
                e.v[GEOAC_EIG_CELERITY] = geo.gc_distance(src[0], src[1], rcv[0], rcv[1]) / travel_time;
                e.v[GEOAC_EIG_AMP_DB] = 20.0 * log10(R[GEOAC_REC_AMP]);
                e.v[GEOAC_EIG_ATTEN_DB] = -attenuation;
                e.v[GEOAC_EIG_INCL] = arrival_incl; e.v[GEOAC_EIG_BEARING] = bearing_back; e.v[GEOAC_EIG_BACKAZ] = back_az; e.v[GEOAC_EIG_AZDEV] = back_az_dev;
                e.smp = fin.smp;
                e.v[GEOAC_EIG_NSMP] = (double)(e.smp.size() / GEOAC_SMP_STRIDE);
                if(verbose){
                    LOG << '\t' << '\t' << "Eigenray-" << DLog::Count{} << ".  " << bnc_cnt << " bounce(s)." << '\n';
                    LOG << '\t' << '\t' << '\t' << "theta, phi = " << DLog::Prec{8} << lt << ", " << 90.0 - lp << " degrees." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Travel Time = " << travel_time << " seconds." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Celerity = " << e.v[GEOAC_EIG_CELERITY] << " km/s." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Amplitude = " << e.v[GEOAC_EIG_AMP_DB] << " dB." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Atmospheric Attenuation = " << -attenuation << " dB." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Arrival inclination = " << arrival_incl << " degrees." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Bearing to source = " << bearing_back << " degrees." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Back azimuth of arrival = " << back_az << " degrees." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Azimuth Deviation = " << back_az_dev << " degrees." << '\n' << '\n';
                } else {
                    LOG << '\t' << "Eigenray identified:" << '\t' << "theta, phi = " << DLog::Prec{8} << lt << ", " << 90.0 - lp << " degrees." << '\n';
                }
                FOUND.push_back(e);
                LOG.incr();
                break;
            } else if(n > 0 && dr > dr_prev){
                lt -= dlt * step_scalar;
                lp -= dlp * step_scalar;
                step_scalar /= 2.0;
                if(sqrt(dlt * dlt + dlp * dlp) * step_scalar < 1.0e-12){
                    if(verbose) LOG << '\t' << '\t' << '\t' << "Step size too small, psuedo-critical ray path likely." << '\n' << '\n';
                    break;
                }
            } else {
                step_scalar = std::min(1.0, step_scalar * 1.25);
                d_lat = rcv[0] * Pi / 180.0 - lat;
                d_lon = rcv[1] * Pi / 180.0 - lon;
                const double rg = geo.r_earth + z_grnd;
                d_lat_dlt = S[7]  - 1.0 / rg * S[4] / S[3] * S[6];
                d_lat_dlp = S[13] - 1.0 / rg * S[4] / S[3] * S[12];
                d_lon_dlt = S[8]  - 1.0 / (rg * cos(lat)) * S[5] / S[3] * S[6];
                d_lon_dlp = S[14] - 1.0 / (rg * cos(lat)) * S[5] / S[3] * S[12];
                det = d_lat_dlt * d_lon_dlp - d_lat_dlp * d_lon_dlt;
                dlt = (d_lon_dlp * d_lat - d_lat_dlp * d_lon) / det * 180.0 / Pi;
                dlp = (-d_lon_dlt * d_lat + d_lat_dlt * d_lon) / det * 180.0 / Pi;
                if(dlt >  lt_lim_step) dlt =  lt_lim_step;
                if(dlt < -lt_lim_step) dlt = -lt_lim_step;
                if(dlp >  lp_lim_step) dlp =  lp_lim_step;
                if(dlp < -lp_lim_step) dlp = -lp_lim_step;
                lt += dlt * step_scalar;
                lp += dlp * step_scalar;
                dr_prev = dr;
            }
        }
    }

    // ---- the text around the -eig_search driver loop: GeoAcGlobal_main.cpp:566-580 ----
    void header(int n_bnc) override { LOG << "Searching for " << n_bnc << " bounce eigenrays." << '\n'; }
    void footer() override { LOG << "Identified " << DLog::Count{} << " eigenray(s)." << '\n'; }
    void run_direct(double theta_est, double phi_from_north, int bounces) override { direct_refine(theta_est, 90.0 - phi_from_north, bounces); }
};

// ================= 3-D Cartesian sets: GeoAc.Eigenray.cpp =================
double modify_d_theta_cart(double dr, double dr_dtheta){       // GeoAc.Eigenray.cpp:24-28
    double width = 2.0 * pow(dr_dtheta, 2);
    return d_theta_big - (d_theta_big - d_theta_small) * exp(-dr * dr / width);
}

struct SearchCart : SearchBase {
    bool strat = true;                     // GeoAc_AtmoStrat: GeoAc3D (12-component rows) vs GeoAc3D.RngDep (18)
    double M_Comps[3] = {0, 0, 0};         // wind Mach numbers at the source (stratified set only, :130-134)

    // ---- GeoAc_EstimateEigenray: GeoAc.Eigenray.cpp:30-121 ----
    bool estimate(double theta_min, double theta_max, double& theta_estimate, double& phi_estimate, double& theta_next, int bounces) override {
        const double azimuth_error_limit = prm.azimuth_err_lim;
        double r_rcvr = sqrt(pow(rcv[0] - src[0], 2) + pow(rcv[1] - src[1], 2));
        double phi = 180.0 / 3.14159 * atan2(rcv[1] - src[1], rcv[0] - src[0]);
        if(verbose){
            LOG << '\t' << "Estimating eigenray angles for source-receiver separated by " << r_rcvr << " km, and azimuth " << 90.0 - phi;
            LOG << " degrees from N.  Inclination limits: [" << theta_min << ", " << theta_max << "]." << '\n';
        }
        int iterations = 0;
        theta_estimate = theta_max;
        double r, r_prev, d_theta = d_theta_big, d_phi = 10.0;
        bool theta_max_reached = false;
        while(fabs(d_phi) > azimuth_error_limit && iterations < 5){
            r = r_rcvr; r_prev = r_rcvr;
            Request scan; scan.bounces = bounces; scan.calc_amp = 0; scan.mode = 0;
            const bool batched = (iterations < 3);
            if(batched){
                for(double theta = theta_min; theta <= theta_max; theta += d_theta){ scan.th.push_back(theta); scan.ph.push_back(90.0 - phi); }
                if(!scan.th.empty() && !trace(scan)) return false;
            }
            int j = 0;
            for(double theta = theta_min; theta <= theta_max; theta += d_theta, j++){
                if(theta + d_theta >= theta_max) theta_max_reached = true;
                Request one; const Request* rq = &scan; int idx = j;
                if(!batched){
                    one.bounces = bounces; one.calc_amp = 0; one.mode = 0; one.th.push_back(theta); one.ph.push_back(90.0 - phi);
                    if(!trace(one)) return false;
                    rq = &one; idx = 0;
                }
                const bool BreakCheck = broke(*rq, idx);
                const double* Rk = last_row(*rq, idx);
                const double xk = Rk[GEOAC_REC_STATE + 0], yk = Rk[GEOAC_REC_STATE + 1];
                if(verbose){
                    LOG << '\t' << '\t' << "Ray launched at " << theta << " degrees arrives at range " << sqrt(pow(xk - src[0], 2) + pow(yk - src[1], 2));
                    LOG << " km after " << bounces << " reflections." << '\t' << "Exact arrival at " << xk << " km East, " << yk << " km North" << '\n';
                }
                if(BreakCheck){ r = r_rcvr; r_prev = r_rcvr; }
                else { r = sqrt(pow(xk - src[0], 2) + pow(yk - src[1], 2)); }
                if((r - r_rcvr) * (r_prev - r_rcvr) < 0.0){
                    if(iterations == 0) theta_next = theta;
                    d_phi = (atan2(rcv[1] - src[1], rcv[0] - src[0]) - atan2(yk - src[1], xk - src[0])) * 180.0 / Pi;
                    while(d_phi > 180.0)  d_phi -= 360.0;
                    while(d_phi < -180.0) d_phi += 360.0;
                    if(fabs(d_phi) < azimuth_error_limit){
                        if(verbose) LOG << '\t' << '\t' << "Azimuth deviation = " << d_phi << ".  Less than " << azimuth_error_limit << " degrees.  Estimates acceptable." << '\n' << '\n';
                        theta_estimate = theta - d_theta;
                        phi_estimate = phi;
                        return true;
                    } else {
                        if(verbose) LOG << '\t' << '\t' << "Azimuth deviation = " << d_phi << ".  Greater than " << azimuth_error_limit << " degrees.  Compensating and searching inclinations again." << '\n' << '\n';
                        phi += d_phi * 0.9;
                        theta_min = std::max(theta - 7.5, theta_min);
                    }
                    break;
                }
                if(iterations >= 3){ d_theta = modify_d_theta_cart(r - r_rcvr, (r - r_prev) / (2.0 * d_theta)); }
                r_prev = r;
            }
            if(theta_max_reached){
                theta_next = theta_max;
                break;
            }
            iterations++;
            if(iterations >= 1 && iterations < 3){ d_theta = d_theta_big / 2.0; }
        }
        if(verbose) LOG << '\t' << '\t' << "Reached maximum inclination angle or iteration limit." << '\n';
        return false;
    }

    // ---- GeoAc_3DEigenray_LM: GeoAc.Eigenray.cpp:123-335 ----
    void refine(double& theta, double& phi, int bnc_cnt, int iterate_limit) override {
        double dr, dr_prev = 10000.0;
        const double tolerance = 0.1;
        const double theta_lim_step = 0.2, phi_lim_step = 0.2;
        double step_scalar = 1.0;
        double nu0[3], M, nu0_xy[2] = {0, 0};
        long double x, y, dx, dy, dx_dt, dy_dt, dx_dp, dy_dp;
        long double det, dt = 0, dp = 0;
        if(verbose) LOG << '\t' << '\t' << "Searching for exact eigenray using auxiliary parameters." << '\n';
        for(int n = 0; n <= iterate_limit; n++){
            if(n == iterate_limit){
                if(verbose){ LOG << '\t' << '\t' << '\t' << "Search for exact eigenray maxed out iterations.  No eigneray idenfied." << '\n' << '\n'; }
                break;
            }
            const double GeoAc_theta = theta * Pi / 180.0, GeoAc_phi = phi * Pi / 180.0;
            if(strat){
                nu0[0] = cos(GeoAc_theta) * cos(GeoAc_phi);
                nu0[1] = cos(GeoAc_theta) * sin(GeoAc_phi);
                nu0[2] = sin(GeoAc_theta);
                M = 1.0 + (nu0[0] * M_Comps[0] + nu0[1] * M_Comps[1] + nu0[2] * M_Comps[2]);
                nu0_xy[0] = nu0[0] / M;
                nu0_xy[1] = nu0[1] / M;
            }
            Request rq; rq.bounces = bnc_cnt; rq.calc_amp = 1; rq.mode = 0;
            rq.th.push_back(theta); rq.ph.push_back(90.0 - phi);
            if(verbose) LOG << '\t' << '\t' << "Plotting ray path with theta = " << theta << ", phi = " << 90.0 - phi;
            if(!trace(rq)) return;
            if(broke(rq, 0)){
                if(verbose) LOG << '\t' << "Ray path left propagation region." << '\n';
                break;
            }
            const double* S = &rq.rec[((size_t)bnc_cnt) * GEOAC_REC_STRIDE + GEOAC_REC_STATE];
            x = S[0]; dx = rcv[0] - x;
            y = S[1]; dy = rcv[1] - y;
            dr = (double)sqrtl(dx * dx + dy * dy);
            if(verbose) LOG << '\t' << '\t' << "Arrival after " << bnc_cnt << " reflections at (" << (double)x << ", " << (double)y << "), distance to receiver = " << dr << " km." << '\n';

            if(dr < tolerance){
                Request fin; fin.bounces = bnc_cnt; fin.calc_amp = 1; fin.mode = GEOAC_MODE_WRITE_RAYS;
                fin.th.push_back(theta); fin.ph.push_back(90.0 - phi);
                if(!trace(fin)) return;
                const double* R = &fin.rec[((size_t)bnc_cnt) * GEOAC_REC_STRIDE];
                const double* Sk = R + GEOAC_REC_STATE;
                Eigenray e; memset(e.v, 0, sizeof e.v);
                const double travel_time = R[GEOAC_REC_TTIME], attenuation = R[GEOAC_REC_ATTEN];
                double back_az = strat ? (90.0 - GeoAc_phi * 180.0 / Pi) + 180.0 : 90.0 - atan2(-Sk[4], -Sk[3]) * 180.0 / Pi;
                double arrival_incl = R[GEOAC_REC_INCL];          // -asin(c(x_k, y_k, z_grnd) / c(src) nu_z) in both mains' records
                double az_to_src = 90.0 - atan2(src[1] - rcv[1], src[0] - rcv[0]) * 180.0 / Pi;
                double back_az_dev = back_az - az_to_src;
                while(back_az > 180.0)      back_az -= 360.0;
                while(back_az < -180.0)     back_az += 360.0;
                while(back_az_dev > 180.0)  back_az_dev -= 360.0;
                while(back_az_dev < -180.0) back_az_dev += 360.0;
                e.v[GEOAC_EIG_RCVR] = rcvr_index; e.v[GEOAC_EIG_INDEX] = 0;  /* numbered when the receiver's segments are put in order */ e.v[GEOAC_EIG_BOUNCES] = bnc_cnt;
                e.v[GEOAC_EIG_THETA] = theta; e.v[GEOAC_EIG_PHI] = 90.0 - phi;
                e.v[GEOAC_EIG_TTIME] = travel_time;
                e.v[GEOAC_EIG_CELERITY] = sqrt(pow(Sk[0] - src[0], 2) + pow(Sk[1] - src[1], 2)) / travel_time;
                e.v[GEOAC_EIG_AMP_DB] = 20.0 * log10(R[GEOAC_REC_AMP]);
                e.v[GEOAC_EIG_ATTEN_DB] = -attenuation;
                e.v[GEOAC_EIG_INCL] = arrival_incl; e.v[GEOAC_EIG_BEARING] = az_to_src; e.v[GEOAC_EIG_BACKAZ] = back_az; e.v[GEOAC_EIG_AZDEV] = back_az_dev;
                e.smp = fin.smp;
                e.v[GEOAC_EIG_NSMP] = (double)(e.smp.size() / GEOAC_SMP_STRIDE);
                if(!verbose) LOG << '\t' << "Eigenray identified:" << '\t' << "theta, phi = " << DLog::Prec{8} << theta << ", " << 90.0 - phi << " degrees." << '\n';
                if(verbose){
                    LOG << '\t' << '\t' << "Eigenray Identified:" << '\n';
                    LOG << '\t' << '\t' << '\t' << "theta, phi = " << DLog::Prec{8} << theta << ", " << 90.0 - phi << " degrees." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Travel Time = " << travel_time << " seconds." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Celerity = " << e.v[GEOAC_EIG_CELERITY] << " km/s." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Amplitude (geometric) = " << e.v[GEOAC_EIG_AMP_DB] << " dB." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Atmospheric Attenuation = " << -attenuation << " dB." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Arrival inclination = " << arrival_incl << " degrees." << '\n';
                    LOG << '\t' << '\t' << '\t' << "Azimuth to source = " << az_to_src << '\n';
                    LOG << '\t' << '\t' << '\t' << "Back Azimuth of arrival = " << back_az << '\n';
                    LOG << '\t' << '\t' << '\t' << "Azimuth Deviation = " << back_az_dev << " degrees." << '\n' << '\n';
                }
                FOUND.push_back(e);
                LOG.incr();
                break;
            } else if(n > 0 && dr > dr_prev){
                theta -= dt * step_scalar;
                phi -= dp * step_scalar;
                step_scalar /= 2.0;
                if(sqrt(dt * dt + dp * dp) * step_scalar < 1.0e-12){
                    if(verbose) LOG << '\t' << '\t' << '\t' << "Step size too small, psuedo-critical ray path likely." << '\n' << '\n';
                    break;
                }
            } else {
                step_scalar = std::min(1.0, step_scalar * 1.25);
                if(strat){
                    dx_dt = S[4] - nu0_xy[0] / S[3] * S[6];
                    dy_dt = S[5] - nu0_xy[1] / S[3] * S[6];
                    dx_dp = S[8] - nu0_xy[0] / S[3] * S[10];
                    dy_dp = S[9] - nu0_xy[1] / S[3] * S[10];
                } else {
                    dx_dt = S[6] - S[3] / S[5] * S[8];
                    dy_dt = S[7] - S[4] / S[5] * S[8];
                    dx_dp = S[12] - S[3] / S[5] * S[14];
                    dy_dp = S[13] - S[4] / S[5] * S[14];
                }
                det = dx_dt * dy_dp - dx_dp * dy_dt;
                dt = 1.0 / det * (dy_dp * dx - dx_dp * dy) * 180.0 / Pi;
                dp = 1.0 / det * (dx_dt * dy - dy_dt * dx) * 180.0 / Pi;
                if(dt > theta_lim_step)  dt =  theta_lim_step;
                if(dp > phi_lim_step)    dp =  phi_lim_step;
                if(dt < -theta_lim_step) dt = -theta_lim_step;
                if(dp < -phi_lim_step)   dp = -phi_lim_step;
                theta += dt * step_scalar;
                phi += dp * step_scalar;
                dr_prev = dr;
            }
        }
    }

    // ---- the text around the -eig_search driver loop: GeoAc3D_main.cpp:531-543 ----
    void header(int n_bnc) override { LOG << "Searching for " << n_bnc << " bounce eigenray(s) between " << prm.theta_min << " and " << prm.theta_max << "." << '\n'; }
    void footer() override { LOG << '\t' << "Identified " << DLog::Count{} << " eigenray(s)." << '\n'; }
    void run_direct(double theta_est, double phi_from_north, int bounces) override { direct_refine(theta_est, 90.0 - phi_from_north, bounces); }
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
    rc = geoac_fan_run(ctx, n, th.data(), ph.data(), rec.data(), &steps);
    if(rc) return rc;
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
static int serve(std::vector<geoac_ctx*>& ctxs, const geoac_params& base, std::vector<Request*>& reqs, geoac_eig_result* res){
    std::map<std::tuple<int, int, int>, std::vector<Request*>> groups;
    for(Request* r : reqs) groups[std::make_tuple(r->bounces, r->calc_amp, r->mode)].push_back(r);
    while(ctxs.size() < groups.size() && ctxs.size() < 8){
        geoac_ctx* c = nullptr;
        int rc = geoac_clone(ctxs[0], &c);
        if(rc) break;                                              // (no clone: the groups take turns on the contexts there are)
        ctxs.push_back(c);
    }
    std::vector<std::pair<const std::tuple<int, int, int>*, std::vector<Request*>*>> work;
    for(auto& g : groups) work.emplace_back(&g.first, &g.second);
    std::atomic<size_t> next{0};
    std::atomic<int> first_rc{0};
    std::mutex mu;
    uint64_t round_crit = 0;
    auto worker = [&](geoac_ctx* c){
        uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
    geoac_eig_result* res = new geoac_eig_result();
    std::vector<geoac_ctx*> ctxs{ ctx };                          // the caller's context and, made on demand, clones of it (serve)
    Shared sh;
    std::vector<std::unique_ptr<SearchBase>> S;
    for(int i = 0; i < n_rcvr; i++){
        std::unique_ptr<SearchBase> sp;
        if(sph){
            Search* q = new Search(); q->geo.r_earth = base.r_earth; sp.reset(q);
            q->src[0] = base.src[1]; q->src[1] = base.src[2]; q->src[2] = std::max(base.src[0], base.z_grnd);   // Source_Loc = (lat, lon, max(z, z_grnd))
        } else {
            SearchCart* q = new SearchCart(); q->strat = (eqset == GEOAC_EQ_3D); sp.reset(q);
            for(int c = 0; c < 3; c++) q->M_Comps[c] = mach[c];
            q->src[0] = base.src[0]; q->src[1] = base.src[1]; q->src[2] = std::max(base.src[2], base.z_grnd);   // Source_Loc = (x, y, max(z, z_grnd))
        }
        sp->sh = &sh; sp->eqset = eqset; sp->rcvr_index = i; sp->z_grnd = base.z_grnd;
        sp->rcv[0] = rcvr[2 * i]; sp->rcv[1] = rcvr[2 * i + 1];
        sp->prm = *ep; sp->verbose = ep->verbose != 0;
        S.push_back(std::move(sp));
    }
    // receivers go through in groups of at most 96 (each receiver is a thread plus one per bounce count plus its refinement tasks, all
    // of them only waiting for rays: the group size keeps the process well under a thousand threads whatever n_rcvr is)
    int err = 0;
    const int group = 96;
    for(int g0 = 0; g0 < n_rcvr && !err; g0 += group){
        const int g1 = std::min(n_rcvr, g0 + group);
        {
            std::unique_lock<std::mutex> lk(sh.mu);
            sh.active = g1 - g0; sh.waiting = 0; sh.pending.clear();
        }
        std::vector<std::thread> threads;
        for(int i = g0; i < g1; i++){
            threads.emplace_back([&, i]{
                SearchBase& s = *S[(size_t)i];
                if(direct) s.run_direct(theta_est[i], phi_est[i], bounces); else s.run_search();
                if(!s.self_released){                              // run_search released this thread before waiting for its tasks
                    std::unique_lock<std::mutex> lk(sh.mu);
                    sh.active--;
                    sh.cv_work.notify_all();
                }
            });
        }
        for(;;){
            std::vector<Request*> batch;
            {
                std::unique_lock<std::mutex> lk(sh.mu);
                sh.cv_work.wait(lk, [&]{ return sh.active == 0 || sh.waiting == sh.active; });
                if(sh.active == 0) break;
                batch.swap(sh.pending);
            }
            err = serve(ctxs, base, batch, res);
            res->stats[3] += 1;
            {
                std::unique_lock<std::mutex> lk(sh.mu);
                if(err) sh.failed = true;
                for(Request* r : batch){ r->done = true; r->error = err; }
                sh.waiting -= (int)batch.size();
                sh.cv_done.notify_all();
            }
        }
        for(auto& t : threads) t.join();
    }
    for(size_t w = 1; w < ctxs.size(); w++) geoac_destroy(ctxs[w]);
    geoac_set_params(ctx, &base);                                  // restore the caller's bounces / calc_amp / mode
    if(err){ delete res; return err; }
    for(int i = 0; i < n_rcvr; i++){
        SearchBase& s = *S[(size_t)i];
        for(Eigenray& e : s.found){
            e.v[GEOAC_EIG_SMP0] = (double)(res->smp.size() / GEOAC_SMP_STRIDE);
            const size_t idx = res->eig.size() / GEOAC_EIG_STRIDE;
            res->eig.insert(res->eig.end(), e.v, e.v + GEOAC_EIG_STRIDE);
            for(size_t q = 0; q + GEOAC_SMP_STRIDE <= e.smp.size(); q += GEOAC_SMP_STRIDE){
                size_t b = res->smp.size();
                res->smp.insert(res->smp.end(), e.smp.begin() + q, e.smp.begin() + q + GEOAC_SMP_STRIDE);
                res->smp[b + GEOAC_SMP_RAY] = (double)idx;
            }
        }
        res->logs.push_back(s.log.str());
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
