// geoac_duo.h - k_rk4_duo: the wave-specialised RK4 kernel of the stratified Global set with amplitudes (included by geoac_kernels.hip).
//
// STATUS (round 3): built, bit-identical to the one-wave kernels (tests/test_gpu_fullsize.py), and measured SLOWER than the two-lane kernel - the
// base wave alone takes 1.91 us per step, but handing 17 doubles per lane and stage to the other wave costs 0.5-0.9 us more (3.0 us against
// 2.56 us then, 2.23 us now; through HBM rows instead of LDS 2.4 us plus the consumer's lag).  It is NOT on the launch plan: option DUO=1
// selects it (diagnostic values 32 / 66 / 160 switch parts of the hand-off off for timing).  Kept because it documents what the hand-off costs.
//
// GeoAc_Propagate_RK4 (GeoAc.Solver.cpp:12-72) integrates 18 equations per ray: the ray itself (r, lat, lon, nu) and two launch-angle
// derivative systems of 6 (Global.cpp:273-367).  The derivative systems read the ray, the ray never reads them.  The time of a fan is the
// serial time of its longest ray (54 130 steps on the metric fan) at one wave per SIMD - every instruction of that wave costs four cycles
// whatever it does - so the step is cut ACROSS WAVES of a workgroup instead of across lanes:
//
//   waves 0, 1  ("base")  one ray per lane: the 6 ray equations, the step-size rule, break / ground checks, reflections, path rows,
//                         the arrival record's ray part.  After every RK4 stage they publish the 17 stage values the derivative systems
//                         need (GlobalStage) in an LDS slot.
//   waves 2, 3  ("aux")   the same 64 rays, lane for lane: both derivative systems (12 equations) from the published stage values; at a
//                         leg end the Jacobian / amplitude of the arrival record and the reflection of their 12 components.
//
// The base wave is what the longest ray waits for, and it no longer carries the derivative systems (EqGlobalPair: ~1 440 instructions per
// step and lane, of which the ray's own right-hand side is computed twice - once per lane of the pair - and each lane still carries one
// system); the aux wave trails it by one message.  A base wave and its aux wave talk through ONE slot of 17 x 64 doubles and two counters
// (published / consumed): LDS executes the DS instructions of a wave in order, so "data, then counter" needs no wait on the producer
// side, and the consumer reads the counter before the data.  The 153 KiB segment table leaves room for exactly that much when its
// records drop the right node (x1 = the next record's x0: 13 doubles per segment; seg_fetch<13>).
//
// Arithmetic: the very functions of the one-wave kernels (global_base, global_derive, global_aux, the RK4 update, EqGlobal's checks,
// reflection and arrival), on the same operands in the same order: records are bit-identical to k_rk4<EqGlobal<true>> / <EqGlobalPair>.
#ifndef GEOAC_DUO_H_
#define GEOAC_DUO_H_

#define GEOAC_DUO_SLOT_BYTES (GEOAC_GSTAGE_W * 64 * 8)              // 8 x (64 lanes x 16 B) + 64 x 8 B
enum { DUO_EV_NONE = 0, DUO_EV_BRK = 1, DUO_EV_FINAL = 2, DUO_EV_REFLECT = 3, DUO_ACT = 16 };

// LDS bytes of a k_rk4_duo workgroup for a profile of nseg segments
static inline size_t geoac_duo_lds_bytes(int nseg){
    const size_t tabn = ((size_t)nseg * 13 + 1 + 1) & ~(size_t)1;                      // 13-wide records + the last right node, even count
    return tabn * sizeof(double) + 2 * GEOAC_DUO_SLOT_BYTES + 4 * sizeof(int);
}

typedef __attribute__((address_space(3))) volatile int geoac_lds_vint;
typedef __attribute__((address_space(3))) geoac_d2 geoac_lds_d2rw;
typedef __attribute__((address_space(3))) double geoac_lds_dbl;

struct DuoPort {                    // LDS byte addresses (explicit address-space-3 accesses: ds_read / ds_write, waits on lgkmcnt only)
    unsigned slot;                  // this pair's message slot
    unsigned seq;                   // messages published (written by the base wave)
    unsigned ack;                   // messages consumed (written by the aux wave)
    int n;                          // this wave's count: published (base) / consumed (aux)
    int lane;
    bool dead;                      // a wait ran out (~1 s: the other wave of the pair is gone): every later wait returns at once, the wave leaves
};
#define GEOAC_DUO_SPIN_MAX (1 << 22)
DEVINL int duo_ctl_load(unsigned a){ return *(geoac_lds_vint*)(size_t)a; }
DEVINL void duo_ctl_store(unsigned a, int v){ *(geoac_lds_vint*)(size_t)a = v; }

// a message in registers: 8 pairs + 1 (stage values in the order of duo_send_stage; header: v[0] = ds, v[1] = code; leg end: v[0..9])
struct DuoMsg { double v[GEOAC_GSTAGE_W]; };

// ---- producer side (base wave) ----
DEVINL void duo_wait_free(DuoPort& pt){                          // the slot is free once everything published has been consumed
    for(int it = 0; !pt.dead && __builtin_amdgcn_readfirstlane(duo_ctl_load(pt.ack)) != pt.n; it++){
        if(it > GEOAC_DUO_SPIN_MAX) pt.dead = true;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
// the consumed count, read early (issued where the caller stands, back by the time duo_wait_free_peeked looks at it): it only grows
// towards pt.n, so "free" seen early is still "free" later
DEVINL int duo_peek_ack(const DuoPort& pt){ return duo_ctl_load(pt.ack); }
DEVINL void duo_wait_free_peeked(DuoPort& pt, int peek){
    if(__builtin_amdgcn_readfirstlane(peek) != pt.n) duo_wait_free(pt);
    asm volatile("" ::: "memory");
}
DEVINL void duo_publish(DuoPort& pt){                            // behind the data stores (a wave's DS instructions execute in order)
    asm volatile("" ::: "memory");
    pt.n++;
    duo_ctl_store(pt.seq, pt.n);
}
DEVINL void duo_put2(DuoPort& pt, int p, double a, double b){
    geoac_d2 v; v.x = a; v.y = b;
    ((geoac_lds_d2rw*)(size_t)pt.slot)[p * 64 + pt.lane] = v;
}
DEVINL void duo_put1(DuoPort& pt, double a){ ((geoac_lds_dbl*)(size_t)(pt.slot + 8 * 64 * 16))[pt.lane] = a; }
DEVINL void duo_put_stage(DuoPort& pt, const GlobalStage& S){
    duo_put2(pt, 0, S.n0, S.n1); duo_put2(pt, 1, S.n2, S.inm); duo_put2(pt, 2, S.cn, S.icg); duo_put2(pt, 3, S.dc, S.du);
    duo_put2(pt, 4, S.dv, S.v);  duo_put2(pt, 5, S.cg2, S.ir); duo_put2(pt, 6, S.ico, S.sth); duo_put2(pt, 7, S.cth, S.H0);
    duo_put1(pt, S.K2);
}

// ---- consumer side (aux wave) ----
DEVINL void duo_wait_msg(DuoPort& pt){
    for(int it = 0; !pt.dead && __builtin_amdgcn_readfirstlane(duo_ctl_load(pt.seq)) == pt.n; it++){
        if(it > GEOAC_DUO_SPIN_MAX) pt.dead = true;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
DEVINL void duo_consumed(DuoPort& pt){                           // behind the data loads (in order again)
    asm volatile("" ::: "memory");
    pt.n++;
    duo_ctl_store(pt.ack, pt.n);
}
DEVINL void duo_read(const DuoPort& pt, DuoMsg& m){
    #pragma unroll
    for(int p = 0; p < 8; p++){
        const geoac_d2 v = ((geoac_lds_d2rw*)(size_t)pt.slot)[p * 64 + pt.lane];
        m.v[2 * p] = v.x; m.v[2 * p + 1] = v.y;
    }
    m.v[16] = ((geoac_lds_dbl*)(size_t)(pt.slot + 8 * 64 * 16))[pt.lane];
}
DEVINL void duo_stage_of(const DuoMsg& m, GlobalStage& S){
    S.n0 = m.v[0]; S.n1 = m.v[1]; S.n2 = m.v[2]; S.inm = m.v[3]; S.cn = m.v[4]; S.icg = m.v[5]; S.dc = m.v[6]; S.du = m.v[7];
    S.dv = m.v[8]; S.v = m.v[9]; S.cg2 = m.v[10]; S.ir = m.v[11]; S.ico = m.v[12]; S.sth = m.v[13]; S.cth = m.v[14]; S.H0 = m.v[15]; S.K2 = m.v[16];
}

// what the aux wave needs of a leg end: the leg's last ray row (arrival record: Jacobian, amplitude) and the scalars of the reflection
// (Global.cpp:140-205: linear intercept, Q1) that the base wave computes from the ray alone
struct DuoLegEnd { double yn[6], dr_k, dr_g, dnu_r_ds, den; };

// ------------------------------------------------------------------------------------------------
// base wave
// ------------------------------------------------------------------------------------------------
// V: 2 = the shipped hand-off (the base wave reads the consumed count early); 32, 64 (| 2): timing diagnostics - the base wave alone without
// any message / messages consumed but nothing computed from them (records NOT valid): what the ray costs, what the hand-off costs
template <int V>
DEVINL void duo_base(const GeoacDevParams& P, const double* tab13, DuoPort& pt, int col, int slot, bool mine, bool done0, unsigned long long& steps_out, bool& done_out){
    using EQ = EqGlobal<true>;
    const size_t np = (size_t)P.n_pad;
    double* st = P.state + (mine ? slot : 0);
    bool done = done0;
    double y[6];
    #pragma unroll
    for(int e = 0; e < 6; e++) y[e] = st[(ST_Y0 + e) * np];
    int k = (int)st[ST_K * np];
    const int k_lim = (int)(P.step_limit - 1);
    int leg = (int)st[ST_LEG * np];
    double hmax = st[ST_HMAX * np];
    RayCtx C; C.ckey = -1; C.kxy = -1;
    C.c0 = st[ST_C0 * np]; C.nu0 = st[ST_NU0 * np];
    #pragma unroll
    for(int q = 0; q < 6; q++) C.a[q] = st[(ST_AUX0 + q) * np];
    EQ::resume(P, C, y);
    int seg = (int)st[ST_SEG * np] * 13;
    seg_fetch<13>(tab13, seg, C.rec);
    int nr = 0, nle = 0;
    unsigned long long steps_here = 0;
    auto put_row = [&](int row, const double* v){
        double* p = P.path + ((size_t)row * 6) * np + col;
        #pragma unroll
        for(int c = 0; c < 6; c++) p[(size_t)c * np] = v[c];
    };
    if(!done) put_row(nr++, y);                                  // carry row: chunk row 0 = current state
    int ev = DUO_EV_NONE;
    DuoLegEnd L;
    #pragma unroll
    for(int e = 0; e < 6; e++) L.yn[e] = 0.0;
    L.dr_k = L.dr_g = L.dnu_r_ds = L.den = 0.0;

    int peek = 0;
    for(;;){
        const bool act = !done && (nr + 2 <= P.s_rows);
        const bool any_act = __any(act);
        double ds = P.ds_min;
        if(act){
            // running turning height (GeoAcGlobal_main.cpp:294) and GeoAc_Set_ds (Global.cpp:210-217)
            const double h = EQ::height(P, y);
            hmax = (hmax < h) ? h : hmax;
            ds = set_ds(EQ::above_ground(P, y), P.ds_min, P.ds_max);
        }
        // header of the step: what the previous step ended with, whether this lane takes the step, its ds
        if(!(V & 32)){
        if(V & 2) duo_wait_free_peeked(pt, peek); else duo_wait_free(pt);
        duo_put2(pt, 0, ds, (double)(ev | (act ? DUO_ACT : 0)));
        duo_publish(pt);
        }
        if(!(V & 32) && __any(ev != DUO_EV_NONE)){
            duo_wait_free(pt);
            duo_put2(pt, 0, L.yn[0], L.yn[1]); duo_put2(pt, 1, L.yn[2], L.yn[3]); duo_put2(pt, 2, L.yn[4], L.yn[5]);
            duo_put2(pt, 3, L.dr_k, L.dr_g); duo_put2(pt, 4, L.dnu_r_ds, L.den);
            duo_publish(pt);
        }
        ev = DUO_EV_NONE;
        if(!any_act || pt.dead) break;

        // ---- the four RK4 stages (Solver.cpp:33-54), the update as in k_rk4 ----
        double dy[6], yt[6], yn[6];
        #pragma unroll
        for(int e = 0; e < 6; e++){ yt[e] = y[e]; yn[e] = y[e]; }
        const double ds_2 = 0.5 * ds, ds_6 = (1.0 / 6.0) * ds, ds_3 = (1.0 / 3.0) * ds;
        auto peek_hook = [&](){ if(V & 2) peek = duo_peek_ack(pt); };
        auto stage_body = [&](int stage, auto rot0){
            GlobalStage S;
            global_base<true, 13, const double*, decltype(peek_hook), decltype(rot0)::value>(tab13, P, seg, C.rec, yt, C.cur[0], C.cur[1], yt[1] - y[1], dy, S, peek_hook, C.rcp0, stage == 0);
            if(!(V & 32)){
                if(V & 2) duo_wait_free_peeked(pt, peek); else duo_wait_free(pt);
                duo_put_stage(pt, S);
                duo_publish(pt);
            }
            if(V & 128){                                          // (timing diagnostic: the stage values to global memory instead - what a base kernel
                double* g = P.contrib + (size_t)(stage * 18) * np + col;   //  whose derivative systems are integrated by a LATER kernel would cost; one slot per stage, rewritten)
                const double v[17] = { S.n0, S.n1, S.n2, S.inm, S.cn, S.icg, S.dc, S.du, S.dv, S.v, S.cg2, S.ir, S.ico, S.sth, S.cth, S.H0, S.K2 };
                #pragma unroll
                for(int f = 0; f < 17; f++) g[(size_t)f * np] = v[f];
            }
            const double wa = (stage == 2) ? ds : ds_2;
            const double wb = (stage == 0 || stage == 3) ? ds_6 : ds_3;
            #pragma unroll
            for(int e = 0; e < 6; e++){
                yn[e] = __builtin_fma(dy[e], wb, yn[e]);
                yt[e] = __builtin_fma(dy[e], wa, y[e]);
            }
        };
        #pragma unroll 1
        for(int stage = 0; stage < 4; stage++) stage_body(stage, std::false_type());

        if(act){
            k++; steps_here++;
            put_row(nr++, yn);
            bool brk, gnd;
            EQ::checks(P, C, y, yn, k, brk, gnd);
            const bool lim = (k >= k_lim);
            if(V & 2) peek = duo_peek_ack(pt);                    // (for the next step's header, ~100 instructions from here)
            if(brk || gnd || lim){
                // ---- leg end: the ray's part of the record (GeoAcGlobal_main.cpp:293-317); the aux wave adds the rest ----
                double* R = P.rec + ((size_t)(P.perm ? P.perm[slot] : slot) * (P.bounces + 1) + leg) * GEOAC_REC_STRIDE;
                R[GEOAC_REC_STEPS] = (double)((lim && !brk && !gnd) ? k + 1 : k);
                P.legend[(size_t)nle * np + col] = nr - 1; nle++;
                if(lim && !brk && !gnd){ atomicOr(&P.counters[2], 1ull); steps_here++; }
                #pragma unroll
                for(int e = 0; e < 6; e++){ R[GEOAC_REC_STATE + e] = yn[e]; L.yn[e] = yn[e]; }
                if(brk){
                    R[GEOAC_REC_BROKE] = 1.0;
                    done = true; ev = DUO_EV_BRK;
                } else {
                    R[GEOAC_REC_VALID] = 1.0;
                    if(lim && !gnd){ const double hl = EQ::height(P, yn); hmax = (hmax < hl) ? hl : hmax; }   // (as k_rk4)
                    R[GEOAC_REC_TURN] = hmax;
                    EqGlobal<false>::arrival(P, C, slot, yn, R);          // inclination, back azimuth, range: the ray alone
                    if(leg >= P.bounces){
                        done = true; ev = DUO_EV_FINAL;
                    } else {
                        // GeoAc_ApproximateIntercept + GeoAc_SetReflectionConditions (Global.cpp:140-205), the ray's six components
                        const double dr_k = yn[0] - y[0];
                        const double dr_g = y[0] - P.ground;
                        double prev[6];
                        #pragma unroll
                        for(int e = 0; e < 6; e++) prev[e] = y[e] + (y[e] - yn[e]) / dr_k * dr_g;
                        Medium mr = medium_at(P, prev[0]);
                        const double c_ref = mr.c;
                        const double dnu_r_ds = -1.0 / c_ref * (C.c0 / c_ref * mr.dc + prev[4] * mr.dv + prev[5] * mr.du
                                                                + c_ref / prev[0] * (prev[4] * prev[4] + prev[5] * prev[5]));
                        #pragma unroll
                        for(int e = 0; e < 6; e++) y[e] = prev[e];
                        y[0] = P.ground;
                        y[3] = -prev[3];
                        L.dr_k = dr_k; L.dr_g = dr_g; L.dnu_r_ds = dnu_r_ds; L.den = c_ref / C.c0 * prev[3];
                        ev = DUO_EV_REFLECT;
                        leg++; k = 0;
                        EQ::restart(P, C, y);
                        put_row(nr++, y);                         // leg-start row
                    }
                }
            } else {
                #pragma unroll
                for(int e = 0; e < 6; e++) y[e] = yn[e];
                EQ::accept(C);
            }
        }
    }

    if(mine && !done0){
        #pragma unroll
        for(int e = 0; e < 6; e++) st[(ST_Y0 + e) * np] = y[e];
        st[ST_K * np] = (double)k; st[ST_LEG * np] = (double)leg; st[ST_DONE * np] = done ? 1.0 : 0.0;
        st[ST_HMAX * np] = hmax; st[ST_SEG * np] = (double)(seg / 13);
        #pragma unroll
        for(int q = 0; q < 6; q++) st[(ST_AUX0 + q) * np] = C.a[q];
        P.nrows[col] = nr; P.nlegend[col] = nle;
    }
    steps_out = steps_here;
    done_out = done;
}

// ------------------------------------------------------------------------------------------------
// aux wave
// ------------------------------------------------------------------------------------------------
template <int V>
DEVINL void duo_aux(const GeoacDevParams& P, DuoPort& pt, int col, int slot, bool mine, bool done0){
    using EQ = EqGlobal<true>;
    if(V & 32) return;                                           // (timing diagnostic: the base wave alone, no messages)
    if(V & 64){                                                  // (timing diagnostic: messages consumed, nothing computed)
        DuoMsg mm;
        for(;;){
            duo_wait_msg(pt); duo_read(pt, mm); duo_consumed(pt);
            const int code = (int)mm.v[1];
            if(__any((code & 15) != DUO_EV_NONE)){ duo_wait_msg(pt); duo_read(pt, mm); duo_consumed(pt); }
            if(!__any((code & DUO_ACT) != 0) || pt.dead) break;
            for(int sgi = 0; sgi < 4; sgi++){ duo_wait_msg(pt); duo_read(pt, mm); duo_consumed(pt); }
        }
        if(mm.v[0] == 1.2345e300) P.counters[31] = 1ull;        // (keeps the reads alive)
        return;
    }
    const size_t np = (size_t)P.n_pad;
    double* st = P.state + (mine ? slot : 0);
    double y[12], yn[12];
    #pragma unroll
    for(int e = 0; e < 12; e++){ y[e] = st[(ST_Y0 + 6 + e) * np]; yn[e] = y[e]; }
    int leg = (int)st[ST_LEG * np];
    RayCtx C; C.ckey = -1; C.kxy = -1;
    C.c0 = st[ST_C0 * np]; C.nu0 = st[ST_NU0 * np];
    bool prev_act = false;
    DuoMsg m;
    auto recv = [&](){ duo_wait_msg(pt); duo_read(pt, m); duo_consumed(pt); };   // the next message into m

    for(;;){
        recv();                                                  // header of the step
        const double ds = m.v[0];
        const int code = (int)m.v[1];
        const int ev = code & 15;
        const bool act = (code & DUO_ACT) != 0;
        DuoLegEnd L;
        if(__any(ev != DUO_EV_NONE)){
            recv();
            #pragma unroll
            for(int e = 0; e < 6; e++) L.yn[e] = m.v[e];
            L.dr_k = m.v[6]; L.dr_g = m.v[7]; L.dnu_r_ds = m.v[8]; L.den = m.v[9];
        }
        // ---- close the previous step ----
        if(prev_act){
            if(ev == DUO_EV_NONE){
                #pragma unroll
                for(int e = 0; e < 12; e++) y[e] = yn[e];
            } else {
                double* R = P.rec + ((size_t)(P.perm ? P.perm[slot] : slot) * (P.bounces + 1) + leg) * GEOAC_REC_STRIDE;
                #pragma unroll
                for(int e = 0; e < 12; e++) R[GEOAC_REC_STATE + 6 + e] = yn[e];
                if(ev != DUO_EV_BRK){
                    double yf[18];
                    #pragma unroll
                    for(int e = 0; e < 6; e++) yf[e] = L.yn[e];
                    #pragma unroll
                    for(int e = 0; e < 12; e++) yf[6 + e] = yn[e];
                    double amp, D;
                    EQ::amp_jac(P, C, slot, yf, amp, D);
                    R[GEOAC_REC_AMP] = amp;
                    R[GEOAC_REC_JACOB] = D;
                }
                if(ev == DUO_EV_REFLECT){
                    double prev[12];
                    #pragma unroll
                    for(int e = 0; e < 12; e++) prev[e] = y[e] + (y[e] - yn[e]) / L.dr_k * L.dr_g;
                    #pragma unroll
                    for(int e = 0; e < 12; e++) y[e] = prev[e];
                    y[0] = -prev[0]; y[6] = -prev[6];
                    y[3] = -prev[3] + 2.0 * L.dnu_r_ds * prev[0] / L.den;
                    y[9] = -prev[9] + 2.0 * L.dnu_r_ds * prev[6] / L.den;
                    leg++;
                }
            }
        }
        prev_act = act;
        if(!__any(act) || pt.dead) break;

        double yt[12];
        #pragma unroll
        for(int e = 0; e < 12; e++){ yt[e] = y[e]; yn[e] = y[e]; }
        const double ds_2 = 0.5 * ds, ds_6 = (1.0 / 6.0) * ds, ds_3 = (1.0 / 3.0) * ds;
        auto stage_body = [&](int stage){
            recv();
            GlobalStage S;
            duo_stage_of(m, S);
            GlobalDerived D;
            global_derive(S, D);
            double dy[12];
            global_aux(S, D, yt, dy);
            global_aux(S, D, yt + 6, dy + 6);
            const double wa = (stage == 2) ? ds : ds_2;
            const double wb = (stage == 0 || stage == 3) ? ds_6 : ds_3;
            #pragma unroll
            for(int e = 0; e < 12; e++){
                yn[e] = __builtin_fma(dy[e], wb, yn[e]);
                yt[e] = __builtin_fma(dy[e], wa, y[e]);
            }
        };
        #pragma unroll 1
        for(int stage = 0; stage < 4; stage++) stage_body(stage);
    }
    if(mine && !done0){
        #pragma unroll
        for(int e = 0; e < 12; e++) st[(ST_Y0 + 6 + e) * np] = y[e];
    }
}

// ------------------------------------------------------------------------------------------------
// the kernel: 256 threads = two ray groups of 64; waves 0, 1 integrate the rays, waves 2, 3 their derivative systems
// ------------------------------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(256, 1) k_rk4_duo(GeoacDevParams P){
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ double lds_tab[];
    if(threadIdx.x == 0) atomicAdd(&P.counters[5], 1ull);       // (k_gate)
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int pr = wave & 1;
    const bool is_base = wave < 2;
    const int col = P.slot_lo + ((int)blockIdx.x * 2 + pr) * 64 + lane;
    const int col_hi = P.colmap ? min(P.slot_hi, *P.n_cols) : P.slot_hi;
    const bool mine = col < col_hi;
    const int slot = (P.colmap && mine) ? P.colmap[col] : col;
    const size_t np = (size_t)P.n_pad;
    const bool done0 = mine ? (P.state[ST_DONE * np + slot] != 0.0) : true;
    if(is_base && mine && done0){ P.nrows[col] = 0; P.nlegend[col] = 0; }
    if(!__syncthreads_or(!done0)) return;
    // ---- the segment table, 13 doubles per record (x0, the cubics of T, u, v) + the last right node ----
    const int n13 = P.nseg * 13;
    for(int q = (int)threadIdx.x; q < n13; q += 256){
        const int k = q / 13, c = q - 13 * k;
        lds_tab[q] = P.seg[(size_t)k * GEOAC_SEGW + (c ? c + 1 : 0)];
    }
    if(threadIdx.x == 0) lds_tab[n13] = P.seg[(size_t)(P.nseg - 1) * GEOAC_SEGW + 1];
    const unsigned slots = (unsigned)(size_t)(geoac_lds_char*)(lds_tab + ((n13 + 2) & ~1));    // LDS byte address of the two message slots
    const unsigned ctl = slots + 2 * GEOAC_DUO_SLOT_BYTES;
    if(threadIdx.x < 4) duo_ctl_store(ctl + 4 * threadIdx.x, 0);
    __syncthreads();
    DuoPort pt;
    pt.slot = slots + pr * GEOAC_DUO_SLOT_BYTES; pt.seq = ctl + 8 * pr; pt.ack = ctl + 8 * pr + 4; pt.n = 0; pt.lane = lane; pt.dead = false;
    if(is_base){
        unsigned long long steps_here = 0; bool done = true;
        duo_base<V>(P, lds_tab, pt, col, slot, mine, done0, steps_here, done);
        // step count and live-ray count: one atomic pair per base wave
        unsigned long long s = 0;
        for(int l = 0; l < 64; l++) s += __shfl(steps_here, l);
        const unsigned long long live = __popcll(__ballot(!done));
        if(lane == 0){
            atomicAdd(&P.counters[0], s);
            atomicAdd(&P.counters[P.live_slot], live);
            if(live) atomicAdd(&P.counters[P.live_slot == 1 ? 4 : 7], 2ull);   // waves that still carry a live ray: this one and its aux wave
        }
    } else {
        duo_aux<V>(P, pt, col, slot, mine, done0);
    }
    if(pt.dead && lane == 0) atomicOr(&P.counters[2], 8ull);    // a hand-off timed out: the host reports the fan as failed
}

#endif  // GEOAC_DUO_H_
