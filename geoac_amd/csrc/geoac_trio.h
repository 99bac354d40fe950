// geoac_trio.h - k_rk4_trio: the wave-specialised RK4 kernel of the stratified Global set with amplitudes (included by geoac_kernels.hip).
//
// GeoAc_Propagate_RK4 (GeoAc.Solver.cpp:12-72) integrates 18 equations per ray: the ray itself (r, lat, lon, nu) and two launch-angle
// derivative systems of 6 (Global.cpp:273-367).  The derivative systems read the ray, the ray never reads them, and the time of a fan is
// the serial time of its longest ray (54 130 steps on the metric fan).  A lone wave on a SIMD issues one instruction every 4.4-6 cycles
// whatever its kind, so the only way to shorten that ray's step is to take instructions OFF its wave - across the SIMDs of a CU, which
// run different instruction streams (lanes of one wave cannot: the two-lane kernel EqGlobalPair still carries the whole ray in both lanes):
//
//   wave 0  ("base")   one ray per lane: the 6 ray equations, the step-size rule, break / ground checks, reflections, path rows, the
//                      ray's part of the leg records - k_rk4's step loop for EqGlobal<false>, lane for lane.  After every RK4 stage it
//                      publishes the 17 stage values the derivative systems need (GlobalStage) in an LDS slot.
//   wave 1, wave 2     the same 64 rays, lane for lane: ONE derivative system each (d/d inclination, d/d azimuth: the same code on
//                      different data) from the published stage values; at a leg end their six components of the record row and of
//                      the reflection.
//
// Round 3 built this with two waves (k_rk4_duo, geoac_duo.h: both systems on one wave, ONE message slot) and measured it slower than the
// two-lane kernel: the wave with twelve equations was the slower of the two and the ray waited for it at every stage.  Here each consumer
// carries less than the producer and the slots form a ring of two: a message is written while the consumers still read the one before.
//
// STATUS (round 4): built, bit-identical to the one-wave kernels on every fan tried (tests/test_gpu_fullsize.py, tools/perf_trio.py), the three waves of a
// workgroup sit on three SIMDs (275 of 275 workgroups, HW_ID) - and NOT faster: 2.8 us per step of the critical ray against the two-lane kernel's 2.10
// (profiles/r04_f_trio.txt).  The ray alone (calc_amp = 0) steps in 1.45 us; the base wave here, alone on its CU and never waiting, in 1.94: handing over
// 17 doubles per lane and stage costs the PRODUCER 18 cycles per ds_write2_b64 (the 1 KiB of a wave's 2 x 8 B go out at 64 B per cycle and the wave issues
// nothing else meanwhile: 0.24 us per step), the second derivatives, the control word and the step size the rest; with the consumers reading beside it (their
// DS instructions share the queue the stores go through) 2.2, and with the acknowledgement read back each stage 2.8.  Moving a launch-angle system's inputs
// between SIMDs costs what computing them again costs - which is what the two lanes of EqGlobalPair do.  A/B builds only (`make AB=1`, option TRIO=1;
// 3 / 5 / 11 / 27: the timing diagnostics behind the figures above, records not valid).
//
// Messages (numbered from 0; message n lives in slot n & 1): STAGE - the 17 stage values of this lane's ray (the step size of the step
// goes through a 64-double array of its own, written with stage 0: the consumers keep it in a register for the step, the next write is
// four messages later); CTRL - at the top of every pass of the base wave's outer loop and once after it: per lane, how the row the last
// vote was about ended (nothing / break / last arrival / reflection, with the four scalars of the reflection the ray's own components
// give) and whether the lane enters the step loop now.  One word holds the published count and the kinds of the last two messages;
// LDS executes the DS instructions of a wave in order, so "data, then word" needs no wait on the producer side and "read data, then
// acknowledge" none on the consumers'.  A consumer learns that the step loop was left (a vote) from the kind of the next message.
//
// Arithmetic: the very functions of the one-wave kernels (global_base, global_derive, global_aux, EqGlobal's checks / restart, the
// reflection) on the same operands in the same order, floating-point contraction off: records are bit-identical to
// k_rk4<EqGlobal<true>> / <EqGlobalPair> (tests/test_gpu_fullsize.py).
#ifndef GEOAC_TRIO_H_
#define GEOAC_TRIO_H_

#define GEOAC_TRIO_SLOT_BYTES (GEOAC_GSTAGE_W * 64 * 8)             // 8 x (64 lanes x 16 B) + 64 x 8 B
enum { TRIO_STAGE = 0, TRIO_CTRL = 1 };
enum { TRIO_EV_NONE = 0, TRIO_EV_BRK = 1, TRIO_EV_FINAL = 2, TRIO_EV_REFLECT = 3, TRIO_IN = 16 };

// LDS bytes of a k_rk4_trio workgroup for a profile of nseg segments: the packed table, two message slots, the step sizes, the control words
static inline size_t geoac_trio_lds_bytes(int nseg){
    const size_t tabn = ((size_t)nseg * 13 + 1 + 1) & ~(size_t)1;                      // 13-wide records + the last right node, even count
    return tabn * sizeof(double) + 2 * GEOAC_TRIO_SLOT_BYTES + 64 * sizeof(double) + 8 * sizeof(int);
}

typedef int geoac_i2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile int trio_lds_vint;
typedef __attribute__((address_space(3))) volatile geoac_i2 trio_lds_vint2;
typedef __attribute__((address_space(3))) double trio_lds_dbl;

// A slot holds the 17 doubles of lane l at byte 136 l (lane-major, an odd number of doubles apart: the 16 lanes of a DS pass fall into different banks; the values of a
// lane sit side by side, so two of them - from any two registers - go out in one ds_write2_b64 and come back in one ds_read2_b64)
struct TrioPort {                   // LDS byte addresses (explicit address-space-3 accesses: ds_read / ds_write, waits on lgkmcnt only)
    unsigned mine;                  // this lane's 136 bytes of slot 0; slot 1: + GEOAC_TRIO_SLOT_BYTES
    unsigned dsv;                   // this lane's entry of ds[64]
    unsigned seq;                   // published count << 2 | kind of the last message << 1 | kind of the one before (written by the base wave)
    unsigned ack;                   // ack[2]: messages consumed by wave 1, wave 2 (8-byte aligned: read as a pair)
    int n;                          // this wave's count: published (base) / consumed (aux)
    int last_kind;                  // base: kind of the message published last
    bool dead;                      // a wait ran out (~1 s: another wave of the workgroup is gone): every later wait returns at once, the wave leaves
};
#define GEOAC_TRIO_SPIN_MAX (1 << 22)
#ifndef GEOAC_TRIO_POLL_SLEEP
#define GEOAC_TRIO_POLL_SLEEP 2          // x 64 cycles between two looks at the word
#endif

// the step loops run with the lanes that take no step switched off: their copies of the port's counts go stale in there.  Behind the loop every
// lane takes the counts of a lane that was in it (`in`: which lanes were; at least one)
DEVINL void trio_port_uniform(TrioPort& pt, bool in){
    const int src = __ffsll((long long)__ballot(in)) - 1;
    pt.n = __builtin_amdgcn_readlane(pt.n, src);
    pt.last_kind = __builtin_amdgcn_readlane(pt.last_kind, src);
    pt.dead = __any(pt.dead);
}
DEVINL trio_lds_dbl* trio_slot(const TrioPort& pt){ return (trio_lds_dbl*)(size_t)(pt.mine + (unsigned)(pt.n & 1) * GEOAC_TRIO_SLOT_BYTES); }   // of message pt.n

// ---- producer side (base wave) ----
DEVINL geoac_i2 trio_peek_acks(const TrioPort& pt){ return *(trio_lds_vint2*)(size_t)pt.ack; }
// slot (pt.n & 1) is free once message pt.n - 2 has been consumed by both: both counts >= pt.n - 1.  `peek`: the counts as read a while
// ago (they only grow: "free" seen early is still free)
DEVINL void trio_wait_free(TrioPort& pt, geoac_i2 peek){
    const int need = pt.n - 1;
    int a = __builtin_amdgcn_readfirstlane(min(peek.x, peek.y));
    if(__builtin_expect(a < need, 0)){
        #pragma nounroll
        for(int it = 0; !pt.dead; it++){
            __builtin_amdgcn_s_sleep(1);
            const geoac_i2 v = trio_peek_acks(pt);
            a = __builtin_amdgcn_readfirstlane(min(v.x, v.y));
            if(a >= need) break;
            if(it > GEOAC_TRIO_SPIN_MAX) pt.dead = true;
        }
    }
    asm volatile("" ::: "memory");
}
DEVINL void trio_publish(TrioPort& pt, int kind){                // behind the data stores (a wave's DS instructions execute in order)
    asm volatile("" ::: "memory");
    pt.n++;
    *(trio_lds_vint*)(size_t)pt.seq = (pt.n << 2) | (kind << 1) | pt.last_kind;
    pt.last_kind = kind;
}
DEVINL void trio_put_stage(const TrioPort& pt, const GlobalStage& S){
    trio_lds_dbl* d = trio_slot(pt);
    d[0] = S.n0; d[1] = S.n1; d[2] = S.n2; d[3] = S.inm; d[4] = S.cn; d[5] = S.icg; d[6] = S.dc; d[7] = S.du; d[8] = S.dv; d[9] = S.v; d[10] = S.cg2;
    d[11] = S.ir; d[12] = S.ico; d[13] = S.sth; d[14] = S.cth; d[15] = S.H0; d[16] = S.K2;
}

// ---- consumer side (aux waves) ----
// A message in registers.  trio_fetch reads the word, THEN the slot of message pt.n and the step size (in that order: DS instructions of a wave execute in order) without
// waiting for anything - the slot may still be in the making; trio_ready looks at the word that came with the data, and only if that says "not published yet" reads again.
// So the usual message costs its consumer one LDS round trip, and that one is issued before the arithmetic on the message before (the ring holds two).
struct TrioMsg { double v[GEOAC_GSTAGE_W]; double ds; int w; };
DEVINL void trio_fetch(const TrioPort& pt, TrioMsg& m){
    m.w = *(trio_lds_vint*)(size_t)pt.seq;
    asm volatile("" ::: "memory");
    const trio_lds_dbl* d = trio_slot(pt);
    #pragma unroll
    for(int f = 0; f < GEOAC_GSTAGE_W; f++) m.v[f] = d[f];
    m.ds = *(trio_lds_dbl*)(size_t)pt.dsv;
    asm volatile("" ::: "memory");
}
// makes m message pt.n (waits for it) and returns its kind; the message stays in its slot until trio_ack
DEVINL int trio_ready(TrioPort& pt, TrioMsg& m){
    int w = __builtin_amdgcn_readfirstlane(m.w);
    if(__builtin_expect((w >> 2) <= pt.n, 0)){
        // (polls the word alone, and not too often: every DS instruction of a waiting wave stands in the queue the base wave's stores go through)
        #pragma nounroll
        for(int it = 0; !pt.dead; it++){
            __builtin_amdgcn_s_sleep(GEOAC_TRIO_POLL_SLEEP);
            w = __builtin_amdgcn_readfirstlane(*(trio_lds_vint*)(size_t)pt.seq);
            if((w >> 2) > pt.n) break;
            if(it > GEOAC_TRIO_SPIN_MAX) pt.dead = true;
        }
        asm volatile("" ::: "memory");
        trio_fetch(pt, m);
    }
    if(pt.dead) return TRIO_CTRL;
    return ((w >> 2) == pt.n + 1) ? ((w >> 1) & 1) : (w & 1);    // (the producer is at most two messages ahead)
}
DEVINL void trio_ack(TrioPort& pt, int q){                       // behind the data loads (in order again)
    asm volatile("" ::: "memory");
    pt.n++;
    ((trio_lds_vint*)(size_t)pt.ack)[q] = pt.n;
}
DEVINL void trio_stage_of(const TrioMsg& m, GlobalStage& S){
    S.n0 = m.v[0]; S.n1 = m.v[1]; S.n2 = m.v[2]; S.inm = m.v[3]; S.cn = m.v[4]; S.icg = m.v[5]; S.dc = m.v[6]; S.du = m.v[7]; S.dv = m.v[8]; S.v = m.v[9]; S.cg2 = m.v[10];
    S.ir = m.v[11]; S.ico = m.v[12]; S.sth = m.v[13]; S.cth = m.v[14]; S.H0 = m.v[15]; S.K2 = m.v[16];
}

// ------------------------------------------------------------------------------------------------
// base wave: k_rk4's stratified step loop for the ray alone (EqGlobal<false>'s checks, restart, reflection; global_base with the stage
// values of the derivative systems switched on)
// ------------------------------------------------------------------------------------------------
// V (diagnostic builds of the hand-off, option TRIO=3 / 5; records NOT valid): 2 - the base wave never waits for the consumers; 4 - the consumers take the messages and compute nothing
template <int V>
DEVINL void trio_base(const GeoacDevParams& P, const double* tab13, TrioPort& pt, int col, int slot, bool mine, bool done0, unsigned long long& steps_out, bool& done_out){
    using EQ = EqGlobal<false>;
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
    const size_t row_stride = (size_t)6 * np;
    double* prow = P.path + col;
    auto put_row = [&](const double* v){
        #pragma unroll
        for(int c = 0; c < 6; c++) prow[(size_t)c * np] = v[c];
        prow += row_stride; nr++;
    };
    if(!done) put_row(y);                                        // carry row: chunk row 0 = current state

    geoac_i2 peek; peek.x = 0; peek.y = 0;
    auto peek_hook = [&](){ if(!(V & 2)) peek = trio_peek_acks(pt); };
    auto send_stage = [&](const GlobalStage& S){
        if(!(V & 2)) trio_wait_free(pt, peek);
        if(V & 16){ trio_lds_dbl* d = trio_slot(pt); d[0] = S.n0 + S.n1 + S.n2 + S.inm + S.cn + S.icg + S.dc + S.du + S.dv + S.v + S.cg2 + S.ir + S.ico + S.sth + S.cth + S.H0 + S.K2; }   // (timing diagnostic: one store)
        else
        trio_put_stage(pt, S);
        trio_publish(pt, TRIO_STAGE);
    };
    int evc = TRIO_EV_NONE;
    double L_dr_k = 0.0, L_dr_g = 0.0, L_dnu = 0.0, L_den = 0.0;
    for(;;){
        const bool in = (nr + 2 <= P.s_rows) && !done;
        {   // CTRL: how the row of the last vote ended for this lane, and whether the lane steps now
            peek = trio_peek_acks(pt);
            if(!(V & 2)) trio_wait_free(pt, peek);
            trio_lds_dbl* d = trio_slot(pt);
            d[0] = (double)(evc | (in ? TRIO_IN : 0)); d[1] = L_dr_k; d[2] = L_dr_g; d[3] = L_dnu; d[4] = L_den;
            trio_publish(pt, TRIO_CTRL);
        }
        evc = TRIO_EV_NONE;
        if(!__any(in) || pt.dead) break;

        bool ev = false, pend = false, brk = false, gnd = false, lim = false;
        double dy[6], ys[6], w6 = 0.0;
        if(in) for(;;){
            hmax = __builtin_fmax(hmax, EQ::height(P, y));       // running turning height (GeoAcGlobal_main.cpp:294)
            const double ds = set_ds(EQ::above_ground(P, y), P.ds_min, P.ds_max);        // GeoAc_Set_ds (Global.cpp:210-217)
            const double ds_2 = 0.5 * ds, ds_6 = (1.0 / 6.0) * ds, ds_3 = (1.0 / 3.0) * ds;
            double yt[6];
            GlobalStage S;
            global_base<true, 13, const double*, decltype(peek_hook), true>(tab13, P, seg, C.rec, y, C.cur[0], C.cur[1], 0.0, dy, S, peek_hook, C.rcp0);
            *(trio_lds_dbl*)(size_t)pt.dsv = ds;
            send_stage(S);
            #pragma unroll
            for(int e = 0; e < 6; e++){ ys[e] = __builtin_fma(dy[e], ds_6, y[e]); yt[e] = __builtin_fma(dy[e], ds_2, y[e]); }
            #pragma unroll 1
            for(int stage = 1; stage < 3; stage++){
                global_base<true, 13, const double*, decltype(peek_hook), false>(tab13, P, seg, C.rec, yt, C.cur[0], C.cur[1], yt[1] - y[1], dy, S, peek_hook, C.rcp0);
                send_stage(S);
                const double wa = (stage == 2) ? ds : ds_2;
                #pragma unroll
                for(int e = 0; e < 6; e++){ ys[e] = __builtin_fma(dy[e], ds_3, ys[e]); yt[e] = __builtin_fma(dy[e], wa, y[e]); }
            }
            global_base<true, 13, const double*, decltype(peek_hook), false>(tab13, P, seg, C.rec, yt, C.cur[0], C.cur[1], yt[1] - y[1], dy, S, peek_hook, C.rcp0);
            send_stage(S);
            double t[6];
            #pragma unroll
            for(int e = 0; e < 6; e++) t[e] = __builtin_fma(dy[e], ds_6, ys[e]);
            EQ::checks(P, C, y, t, k + 1, brk, gnd);
            lim = (k + 1 >= k_lim);
            ev = brk || gnd || lim;
            const bool full = !(nr + 3 <= P.s_rows);
            if(__builtin_expect(__any(ev || full), 0)){ pend = true; w6 = ds_6; break; }
            #pragma unroll
            for(int e = 0; e < 6; e++) y[e] = __builtin_fma(dy[e], ds_6, ys[e]);
            k++; steps_here++;
            put_row(y);
            EQ::accept(C);
        }
        trio_port_uniform(pt, in);
        if(pend){                                                 // (rare) the row that was voted on
            #pragma unroll
            for(int e = 0; e < 6; e++) ys[e] = __builtin_fma(dy[e], w6, ys[e]);
            k++; steps_here++;
            put_row(ys);
            EQ::accept(C);
            if(ev){
                // ---- leg end: the ray's part of the record (GeoAcGlobal_main.cpp:293-317); the other waves add their rows, k_arrival the rest ----
                double* R = P.rec + ((size_t)(P.perm ? P.perm[slot] : slot) * (P.bounces + 1) + leg) * GEOAC_REC_STRIDE;
                R[GEOAC_REC_STEPS] = (double)((lim && !brk && !gnd) ? k + 1 : k);
                P.legend[(size_t)nle * np + col] = nr - 1; nle++;
                if(lim && !brk && !gnd){ atomicOr(&P.counters[2], 1ull); steps_here++; }
                #pragma unroll
                for(int e = 0; e < 6; e++) R[GEOAC_REC_STATE + e] = ys[e];
                if(brk){
                    R[GEOAC_REC_BROKE] = 1.0;
                    done = true; evc = TRIO_EV_BRK;
                } else {
                    R[GEOAC_REC_VALID] = 1.0;
                    if(lim && !gnd){ const double hl = EQ::height(P, ys); hmax = (hmax < hl) ? hl : hmax; }
                    R[GEOAC_REC_TURN] = hmax;
                    if(leg >= P.bounces){
                        done = true; evc = TRIO_EV_FINAL;
                    } else {
                        // GeoAc_ApproximateIntercept + GeoAc_SetReflectionConditions (Global.cpp:140-205, Q1), the ray's six components (EqGlobal::reflect)
                        const double dr_k = ys[0] - y[0];
                        const double dr_g = y[0] - P.ground;
                        double prev[6];
                        #pragma unroll
                        for(int e = 0; e < 6; e++) prev[e] = y[e] + (y[e] - ys[e]) / dr_k * dr_g;
                        Medium mr = medium_at(P, prev[0]);
                        const double c_ref = mr.c;
                        const double dnu_r_ds = -1.0 / c_ref * (C.c0 / c_ref * mr.dc + prev[4] * mr.dv + prev[5] * mr.du
                                                                + c_ref / prev[0] * (prev[4] * prev[4] + prev[5] * prev[5]));
                        #pragma unroll
                        for(int e = 0; e < 6; e++) y[e] = prev[e];
                        y[0] = P.ground;
                        y[3] = -prev[3];
                        L_dr_k = dr_k; L_dr_g = dr_g; L_dnu = dnu_r_ds; L_den = c_ref / C.c0 * prev[3];
                        evc = TRIO_EV_REFLECT;
                        leg++; k = 0;
                        EQ::restart(P, C, y);
                        put_row(y);                               // leg-start row
                    }
                }
            } else {
                #pragma unroll
                for(int e = 0; e < 6; e++) y[e] = ys[e];
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
// aux wave q (0: d / d inclination, 1: d / d azimuth)
// ------------------------------------------------------------------------------------------------
template <int V>
DEVINL void trio_aux(const GeoacDevParams& P, TrioPort& pt, int q, int col, int slot, bool mine, bool done0){
    const size_t np = (size_t)P.n_pad;
    double* st = P.state + (mine ? slot : 0);
    double y[6], yn[6];
    #pragma unroll
    for(int e = 0; e < 6; e++){ y[e] = st[(ST_Y0 + 6 + 6 * q + e) * np]; yn[e] = y[e]; }
    int leg = (int)st[ST_LEG * np];
    bool stepped = false;                                        // this lane took part in the step loop that the last vote ended: yn is its new row
    if(V & 8) return;                                            // (timing diagnostic: the base wave alone)
    TrioMsg m, nx;
    // one stage: the message in m is taken (its slot handed back), the next one is sent for, then the arithmetic
    auto stage = [&](const double* ya, double* dy, TrioMsg& next){
        GlobalStage S; GlobalDerived D;
        trio_stage_of(m, S);
        trio_ack(pt, q);
        trio_fetch(pt, next);
        if(V & 4){ for(int e = 0; e < 6; e++) dy[e] = S.n0; return; }
        global_derive(S, D);
        global_aux(S, D, ya, dy);
    };
    for(;;){
        // ---- CTRL (every lane) ----
        trio_fetch(pt, m);
        (void)trio_ready(pt, m);
        trio_ack(pt, q);
        const int code = pt.dead ? 0 : (int)m.v[0];
        const double dr_k = m.v[1], dr_g = m.v[2], dnu = m.v[3], den = m.v[4];
        const int evk = code & 15;
        const bool in = (code & TRIO_IN) != 0;
        if(stepped){
            if(evk == TRIO_EV_NONE){
                #pragma unroll
                for(int e = 0; e < 6; e++) y[e] = yn[e];
            } else {
                double* R = P.rec + ((size_t)(P.perm ? P.perm[slot] : slot) * (P.bounces + 1) + leg) * GEOAC_REC_STRIDE;
                #pragma unroll
                for(int e = 0; e < 6; e++) R[GEOAC_REC_STATE + 6 + 6 * q + e] = yn[e];
                if(evk == TRIO_EV_REFLECT){
                    double prev[6];
                    #pragma unroll
                    for(int e = 0; e < 6; e++) prev[e] = y[e] + (y[e] - yn[e]) / dr_k * dr_g;
                    #pragma unroll
                    for(int e = 0; e < 6; e++) y[e] = prev[e];
                    y[0] = -prev[0];
                    y[3] = -prev[3] + 2.0 * dnu * prev[0] / den;
                    leg++;
                }
            }
        }
        stepped = in;
        if(!__any(in) || pt.dead) break;

        if(in){
            trio_fetch(pt, m);
            int kind = trio_ready(pt, m);                         // stage 0 of the first step
            for(;;){
                if(kind != TRIO_STAGE) break;                     // (only when a wait ran out)
                const double ds = m.ds;
                const double ds_2 = 0.5 * ds, ds_6 = (1.0 / 6.0) * ds, ds_3 = (1.0 / 3.0) * ds;
                double dy[6], yt[6], ys[6];
                stage(y, dy, nx);
                #pragma unroll
                for(int e = 0; e < 6; e++){ ys[e] = __builtin_fma(dy[e], ds_6, y[e]); yt[e] = __builtin_fma(dy[e], ds_2, y[e]); }
                m = nx; (void)trio_ready(pt, m);
                stage(yt, dy, nx);
                #pragma unroll
                for(int e = 0; e < 6; e++){ ys[e] = __builtin_fma(dy[e], ds_3, ys[e]); yt[e] = __builtin_fma(dy[e], ds_2, y[e]); }
                m = nx; (void)trio_ready(pt, m);
                stage(yt, dy, nx);
                #pragma unroll
                for(int e = 0; e < 6; e++){ ys[e] = __builtin_fma(dy[e], ds_3, ys[e]); yt[e] = __builtin_fma(dy[e], ds, y[e]); }
                m = nx; (void)trio_ready(pt, m);
                stage(yt, dy, m);                                 // (the values of m are in hand by the time the next message is sent for)
                #pragma unroll
                for(int e = 0; e < 6; e++) yn[e] = __builtin_fma(dy[e], ds_6, ys[e]);
                kind = trio_ready(pt, m);                 // another step (its stage 0), or the vote's CTRL (left in its slot: every lane reads it above)
                if(kind != TRIO_STAGE) break;
                #pragma unroll
                for(int e = 0; e < 6; e++) y[e] = yn[e];
            }
        }
        trio_port_uniform(pt, in);
    }
    if(mine && !done0){
        #pragma unroll
        for(int e = 0; e < 6; e++) st[(ST_Y0 + 6 + 6 * q + e) * np] = y[e];
    }
}

// ------------------------------------------------------------------------------------------------
// the kernel: 192 threads = 64 rays; wave 0 integrates the rays, waves 1 and 2 one derivative system each
// ------------------------------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(192, 1) k_rk4_trio(GeoacDevParams P){
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ double lds_tab[];
    if(threadIdx.x == 0) atomicAdd(&P.counters[5], 1ull);       // (k_gate)
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int col = P.slot_lo + (int)blockIdx.x * 64 + lane;
    const int col_hi = P.colmap ? min(P.slot_hi, *P.n_cols) : P.slot_hi;
    const bool mine = col < col_hi;
    const int slot = (P.colmap && mine) ? P.colmap[col] : col;
    const size_t np = (size_t)P.n_pad;
    const bool done0 = mine ? (P.state[ST_DONE * np + slot] != 0.0) : true;
    if(wave == 0 && mine && done0){ P.nrows[col] = 0; P.nlegend[col] = 0; }
    if(!__syncthreads_or(!done0)) return;
    // ---- the segment table, 13 doubles per record (x0, the cubics of T, u, v) + the last right node ----
    const int n13 = P.nseg * 13;
    for(int q = (int)threadIdx.x; q < n13; q += 192){
        const int k = q / 13, c = q - 13 * k;
        lds_tab[q] = P.seg[(size_t)k * GEOAC_SEGW + (c ? c + 1 : 0)];
    }
    if(threadIdx.x == 0) lds_tab[n13] = P.seg[(size_t)(P.nseg - 1) * GEOAC_SEGW + 1];
    const unsigned slots = (unsigned)(size_t)(geoac_lds_char*)(lds_tab + ((n13 + 2) & ~1));    // LDS byte address of the two message slots
    const unsigned dsv = slots + 2 * GEOAC_TRIO_SLOT_BYTES;
    const unsigned ctl = dsv + 64 * 8;
    if(threadIdx.x < 4) ((trio_lds_vint*)(size_t)ctl)[threadIdx.x] = 0;
    // (diagnostic, TRACE_EPOCHS: do the three waves sit on three SIMDs?  HW_ID bits 5:4)
    if(lane == 0) ((trio_lds_vint*)(size_t)ctl)[4 + wave] = (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4));
    __syncthreads();
    if(threadIdx.x == 0){
        const int a = ((trio_lds_vint*)(size_t)ctl)[4], b = ((trio_lds_vint*)(size_t)ctl)[5], c = ((trio_lds_vint*)(size_t)ctl)[6];
        atomicAdd(&P.counters[30], (a != b && a != c && b != c) ? 1ull : (1ull << 32));
    }
    TrioPort pt;
    pt.mine = slots + (unsigned)lane * (GEOAC_GSTAGE_W * 8); pt.dsv = dsv + (unsigned)lane * 8; pt.seq = ctl; pt.ack = ctl + 8; pt.n = 0; pt.last_kind = 0; pt.dead = false;
    if(wave == 0){
        unsigned long long steps_here = 0; bool done = true;
        trio_base<V>(P, lds_tab, pt, col, slot, mine, done0, steps_here, done);
        // step count and live-ray count: one atomic pair per base wave
        unsigned long long s = 0;
        for(int l = 0; l < 64; l++) s += __shfl(steps_here, l);
        const unsigned long long live = __popcll(__ballot(!done));
        if(lane == 0){
            atomicAdd(&P.counters[0], s);
            atomicAdd(&P.counters[P.live_slot], live);
            if(live) atomicAdd(&P.counters[P.live_slot == 1 ? 4 : 7], 2ull);   // (in the two-lane kernel's unit: 64 rays are two of its waves)
        }
    } else {
        trio_aux<V>(P, pt, wave - 1, col, slot, mine, done0);
    }
    if(pt.dead && lane == 0) atomicOr(&P.counters[2], 8ull);    // a hand-off timed out: the host reports the fan as failed
}

#endif  // GEOAC_TRIO_H_
