// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// extern "C" surface of the CPU restatement (loaded with ctypes by oracle/oracle.py).  PARITY UNPINNED: the
// reference (/root/reference) holds no golden vectors for this path and cannot be executed here (no Julia);
// see DESIGN.md.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <chrono>
#include <cstring>
#include <memory>
#include <thread>
#include "qp.hpp"

using namespace po;

namespace {

struct Inst {
    bool solved = false;
    std::vector<double> prev_q, prev_u, ts, dt, prev_ts;
    OSQPPort osqp;
    bool osqp_init = false;
};

struct Handle {
    CoupledMPC mpc;
    CoupledQPLayout lay;
    KKTPattern kkt;
    OSQPSettings osqp_settings;
    std::vector<std::unique_ptr<Inst>> inst;
    int sd_len() const { return 84 * mpc.N() + 11; }
};

void sd_to_flat(const StageData& sd, double* f) {
    int N = sd.Ns + sd.Nl; double* p = f;
    auto cp = [&](const std::vector<double>& v) { std::memcpy(p, v.data(), v.size() * sizeof(double)); p += v.size(); };
    cp(sd.A); cp(sd.B0); cp(sd.Bf); cp(sd.c); cp(sd.H); cp(sd.G); cp(sd.dmin); cp(sd.dmax); cp(sd.fxmax); cp(sd.ddmin); cp(sd.ddmax); cp(sd.dt);
    std::memcpy(p, sd.q_curr, 48); p += 6; std::memcpy(p, sd.u_curr, 16); p += 2; std::memcpy(p, sd.M_hji, 16); p += 2; *p++ = sd.b_hji;
    (void)N;
}
void flat_to_sd(const double* f, int Ns, int Nl, StageData& sd) {
    sd.resize(Ns, Nl); const double* p = f;
    auto cp = [&](std::vector<double>& v) { std::memcpy(v.data(), p, v.size() * sizeof(double)); p += v.size(); };
    cp(sd.A); cp(sd.B0); cp(sd.Bf); cp(sd.c); cp(sd.H); cp(sd.G); cp(sd.dmin); cp(sd.dmax); cp(sd.fxmax); cp(sd.ddmin); cp(sd.ddmax); cp(sd.dt);
    std::memcpy(sd.q_curr, p, 48); p += 6; std::memcpy(sd.u_curr, p, 16); p += 2; std::memcpy(sd.M_hji, p, 16); p += 2; sd.b_hji = *p++;
}

}  // namespace

extern "C" {

void* po_create(int Ns, int Nl, double dt_short, double dt_long, int use_correction_step, int rk4_substeps) {
    Handle* h = new Handle();
    h->mpc.init(Ns, Nl, dt_short, dt_long, use_correction_step != 0);
    h->mpc.rk4_substeps = rk4_substeps;
    h->lay.build(Ns, Nl);
    h->kkt.build(h->lay.n, h->lay.m, h->lay.Ap, h->lay.Ai);
    return h;
}
void po_destroy(void* hv) { delete (Handle*)hv; }

// vehicle (22 doubles, order of VehicleParams) and control params (16 doubles, order of CoupledControlParams with N_HJI as double)
void po_get_params(void* hv, double* veh22, double* cp16, double* u_norm2) {
    Handle* h = (Handle*)hv;
    std::memcpy(veh22, &h->mpc.veh, 22 * sizeof(double));
    const CoupledControlParams& c = h->mpc.cp;
    double v[16] = {c.V_min, c.V_max, c.k_V, c.k_s, c.deltadot_max, c.Q_ds, c.Q_dpsi, c.Q_e, c.W_beta, c.W_r, c.W_HJI, (double)c.N_HJI, c.R_delta, c.R_ddelta, c.R_Fx, c.R_dFx};
    std::memcpy(cp16, v, sizeof(v));
    u_norm2[0] = h->mpc.u_norm[0]; u_norm2[1] = h->mpc.u_norm[1];
}
void po_set_control_params(void* hv, const double* v) {
    CoupledControlParams& c = ((Handle*)hv)->mpc.cp;
    c.V_min = v[0]; c.V_max = v[1]; c.k_V = v[2]; c.k_s = v[3]; c.deltadot_max = v[4]; c.Q_ds = v[5]; c.Q_dpsi = v[6]; c.Q_e = v[7];
    c.W_beta = v[8]; c.W_r = v[9]; c.W_HJI = v[10]; c.N_HJI = (int)v[11]; c.R_delta = v[12]; c.R_ddelta = v[13]; c.R_Fx = v[14]; c.R_dFx = v[15];
}
void po_set_hji_eps(void* hv, double eps) { ((Handle*)hv)->mpc.HJI_eps = eps; }

// arrays: [12][L] contiguous in the field order of TrajectoryTube (t,s,V,A,E,N,psi,kappa,theta,phi,edge_L,edge_R)
void po_set_trajectory(void* hv, int L, const double* a) {
    TrajectoryTube& T = ((Handle*)hv)->mpc.traj;
    T.L = L;
    std::vector<double>* f[12] = {&T.t, &T.s, &T.V, &T.A, &T.E, &T.N, &T.psi, &T.kappa, &T.theta, &T.phi, &T.edge_L, &T.edge_R};
    for (int k = 0; k < 12; k++) f[k]->assign(a + (size_t)k * L, a + (size_t)(k + 1) * L);
}
void po_set_hji_grid(void* hv, const int* dims, const float* knots_concat, const float* V, const float* gradV) {
    HJICache& C = ((Handle*)hv)->mpc.hji;
    size_t n = 1; const float* k = knots_concat;
    for (int d = 0; d < 7; d++) { C.dims[d] = dims[d]; C.knots[d].assign(k, k + dims[d]); k += dims[d]; n *= dims[d]; }
    C.V.assign(V, V + n); C.gradV.assign(gradV, gradV + 7 * n); C.loaded = true;
}

// ---- piecewise entry points (unit tests) ---------------------------------------------------------------------
void po_time_steps(void* hv, double t0, double* ts, double* dt) {
    Handle* h = (Handle*)hv; MPCTimeSteps T = h->mpc.TS; T.compute(t0);
    std::memcpy(ts, T.ts.data(), T.ts.size() * 8); std::memcpy(dt, T.dt.data(), T.dt.size() * 8);
}
void po_set_time_grid_naive(void* hv, int naive) { ((Handle*)hv)->mpc.TS.naive_time_grid = naive != 0; }      // A/B: the two-rounding grid of rounds 1-5
// the values `t` takes in `for t in 0:dt:t_end` (model_predictive_control.jl:87), shifted by t_start: out[k] = (t_start .+ (0:dt:t_end))[k + 1]; returns the range's length
int po_simulate_times(double dt, double t_end, double t_start, int steps, double* out) {
    const jlrange::Range r = jlrange::shifted(jlrange::colon(0.0, dt, t_end), t_start);
    for (int k = 0; k < steps; k++) out[k] = jlrange::elem(r, k + 1);
    return (int)r.len;
}
void po_path_coordinates(void* hv, double E, double N, double* out3, int* imin) {
    ((Handle*)hv)->mpc.traj.path_coordinates(E, N, out3[0], out3[1], out3[2], imin);
}
void po_traj_at_time(void* hv, double t, double* out12) { TrajectoryNode n = ((Handle*)hv)->mpc.traj.at_time(t); std::memcpy(out12, &n, 96); }
void po_traj_at_s(void* hv, double s, double* out12) { TrajectoryNode n = ((Handle*)hv)->mpc.traj.at_s(s); std::memcpy(out12, &n, 96); }
void po_tracking_dynamics(void* hv, const double* q, const double* u, const double* p, double* out) { vehicle_tracking_dynamics<double>(((Handle*)hv)->mpc.veh, q, u, p, out); }
void po_world_dynamics(void* hv, const double* q, const double* u, double* out) { vehicle_world_dynamics<double>(((Handle*)hv)->mpc.veh, q, u, out); }
void po_stable_limits(void* hv, double Ux, double Fxf, double Fxr, double* out14) {
    StableLimits s = stable_limits(((Handle*)hv)->mpc.veh, Ux, Fxf, Fxr);
    out14[0] = s.delta_min; out14[1] = s.delta_max; std::memcpy(out14 + 2, s.H, 64); std::memcpy(out14 + 10, s.G, 32);
}
void po_steady_state(void* hv, double V, double A_tan, double kappa, int num_iters, double r, double beta0, double delta0, double Fyf0, double* out8) {
    SteadyState s = steady_state_estimates(((Handle*)hv)->mpc.veh, V, A_tan, kappa, num_iters, r, beta0, delta0, Fyf0);
    std::memcpy(out8, &s, 64);
}
void po_lateral_tire_forces(void* hv, double Ux, double Uy, double r, double delta, double Fxf, double Fxr, double* out2) {
    lateral_tire_forces_q(((Handle*)hv)->mpc.veh, Ux, Uy, r, delta, Fxf, Fxr, out2[0], out2[1]);
}
// B0/Bf NOT normalised here
void po_linearize_interval(void* hv, const double* q, const double* u0, const double* p0, const double* uf, const double* pf, double dt, int ramp,
                           double* A, double* B0, double* Bf, double* c) {
    ((Handle*)hv)->mpc.linearize_interval(q, u0, p0, uf, pf, dt, ramp != 0, A, B0, Bf, c);
}
void po_propagate_tracking(void* hv, double* q, const double* u0, const double* p0, const double* uf, const double* pf, double dt, int ramp) {
    ((Handle*)hv)->mpc.rk4_tracking<double>(q, u0, uf, p0, ramp ? pf : p0, dt, ramp != 0);
}
void po_plant_step(void* hv, double* q6, const double* u3, double dt) { ((Handle*)hv)->mpc.plant_step(q6, u3, dt); }
void po_next_control(void* hv, const double* u2n, double* out3) { ((Handle*)hv)->mpc.next_control(u2n, out3); }

void po_hji_relative_state(const double* us6, const double* them4, double* x7) { hji_relative_state(us6, them4, x7); }
int po_hji_lookup(void* hv, const double* x7, double* V, double* g7) { return ((Handle*)hv)->mpc.hji.lookup(x7, *V, g7) ? 1 : 0; }
// ros_integration.jl:114-124 selection input: (delta_opt, Fx_opt) of optimal_control at the looked-up gradient; returns V
double po_hji_optimal_control(void* hv, const double* state6, const double* other4, double* u2) {
    Handle* h = (Handle*)hv; double x7[7], g[7], V; hji_relative_state(state6, other4, x7);
    h->mpc.hji.lookup(x7, V, g);
    optimal_control(h->mpc.veh, x7, g, u2);
    return V;
}
void po_hji_constraint(void* hv, const double* state6, const double* other4, const double* control3, double* M2, double* b, double* V) {
    Handle* h = (Handle*)hv; double x7[7]; hji_relative_state(state6, other4, x7);
    double uR[2] = {control3[0], control3[1] + control3[2]};
    reachability_constraint(h->mpc.veh, h->mpc.hji, x7, h->mpc.HJI_eps, uR, M2, *b, *V);
}

void po_nodes(void* hv, const double* state6, const double* control3, double time_offset, int solved,
              const double* ts, const double* dt, const double* prev_ts, const double* prev_q, const double* prev_u,
              double* qs, double* us, double* ps) {
    Handle* h = (Handle*)hv; CoupledMPC m = h->mpc;     // copy: TS is scratch
    int Nn = m.N() + 1;
    m.TS.ts.assign(ts, ts + Nn); m.TS.dt.assign(dt, dt + Nn - 1);
    if (prev_ts) m.TS.prev_ts.assign(prev_ts, prev_ts + Nn);
    Nodes nd; m.linearization_nodes(state6, control3, time_offset, solved != 0, prev_q, prev_u, nd);
    std::memcpy(qs, nd.qs.data(), nd.qs.size() * 8); std::memcpy(us, nd.us.data(), nd.us.size() * 8); std::memcpy(ps, nd.ps.data(), nd.ps.size() * 8);
}
int po_sd_len(void* hv) { return ((Handle*)hv)->sd_len(); }
void po_update_qp(void* hv, const double* qs, const double* us, const double* ps, const double* dt,
                  const double* state6, const double* control3, const double* other4, double* sd_flat, double* V_hji) {
    Handle* h = (Handle*)hv; CoupledMPC& m = h->mpc; int Nn = m.N() + 1;
    MPCTimeSteps saved = m.TS; m.TS.dt.assign(dt, dt + Nn - 1);
    Nodes nd; nd.qs.assign(qs, qs + 6 * Nn); nd.us.assign(us, us + 2 * Nn); nd.ps.assign(ps, ps + 4 * Nn);
    StageData sd; m.update_qp(nd, state6, control3, other4, sd, V_hji);
    m.TS = saved;
    sd_to_flat(sd, sd_flat);
}
void po_qp_dims(void* hv, int* n, int* m, int* nnz) { Handle* h = (Handle*)hv; *n = h->lay.n; *m = h->lay.m; *nnz = (int)h->lay.Ai.size(); }
void po_assemble_qp(void* hv, const double* sd_flat, double* Pd, double* q, int* Ap, int* Ai, double* Ax, double* l, double* u) {
    Handle* h = (Handle*)hv; StageData sd; flat_to_sd(sd_flat, h->mpc.TS.N_short, h->mpc.TS.N_long, sd);
    QP qp; h->lay.fill(sd, h->mpc.cp, h->mpc.veh, h->mpc.u_norm, qp);
    std::memcpy(Pd, qp.Pd.data(), qp.n * 8); std::memcpy(q, qp.q.data(), qp.n * 8);
    std::memcpy(Ap, qp.Ap.data(), (qp.n + 1) * 4); std::memcpy(Ai, qp.Ai.data(), qp.Ai.size() * 4); std::memcpy(Ax, qp.Ax.data(), qp.Ax.size() * 8);
    std::memcpy(l, qp.l.data(), qp.m * 8); std::memcpy(u, qp.u.data(), qp.m * 8);
}
// info: [iters, status, res_pri, res_dua, gap, polished (0 = interior-point answer, k = active-set polish verified in round k)]
int po_solve_exact(void* hv, const double* sd_flat, double* x, double* y, double* info5) {
    Handle* h = (Handle*)hv; StageData sd; flat_to_sd(sd_flat, h->mpc.TS.N_short, h->mpc.TS.N_long, sd);
    QP qp; h->lay.fill(sd, h->mpc.cp, h->mpc.veh, h->mpc.u_norm, qp);
    LDLNumeric ldl; ldl.init(&h->kkt.sym);
    ExactResult R; int st = solve_exact_robust(qp, h->kkt, ldl, R);
    if ((int)R.x.size() == qp.n) { std::memcpy(x, R.x.data(), qp.n * 8); std::memcpy(y, R.y.data(), qp.m * 8); }
    info5[0] = R.iters; info5[1] = R.status; info5[2] = R.res_pri; info5[3] = R.res_dua; info5[4] = R.gap; info5[5] = R.polished;
    return st;
}
void po_osqp_settings(void* hv, double rho, double sigma, double alpha, double eps_abs, double eps_rel, int max_iter, int scaling,
                      int check_termination, int adaptive_rho, int adaptive_rho_interval, int warm_start) {
    OSQPSettings& s = ((Handle*)hv)->osqp_settings;
    s.rho = rho; s.sigma = sigma; s.alpha = alpha; s.eps_abs = eps_abs; s.eps_rel = eps_rel; s.max_iter = max_iter; s.scaling = scaling;
    s.check_termination = check_termination; s.adaptive_rho = adaptive_rho; s.adaptive_rho_interval = adaptive_rho_interval; s.warm_start = warm_start;
    for (auto& i : ((Handle*)hv)->inst) if (i) i->osqp_init = false;
}
static Inst& get_inst(Handle* h, int id) {
    if ((int)h->inst.size() <= id) h->inst.resize(id + 1);
    if (!h->inst[id]) h->inst[id].reset(new Inst());
    Inst& I = *h->inst[id];
    if (!I.osqp_init) { I.osqp.init(&h->kkt, h->osqp_settings); I.osqp_init = true; }
    return I;
}
void po_reserve_instances(void* hv, int B) { Handle* h = (Handle*)hv; for (int i = 0; i < B; i++) get_inst(h, i); }
void po_reset_instance(void* hv, int id) { Inst& I = get_inst((Handle*)hv, id); I.solved = false; I.osqp.reset(); }
// info: [iters, status, pri_res, dua_res, rho, n_refactor]
int po_osqp_solve(void* hv, int id, const double* sd_flat, double* x, double* y, double* info6) {
    Handle* h = (Handle*)hv; Inst& I = get_inst(h, id);
    StageData sd; flat_to_sd(sd_flat, h->mpc.TS.N_short, h->mpc.TS.N_long, sd);
    QP qp; h->lay.fill(sd, h->mpc.cp, h->mpc.veh, h->mpc.u_norm, qp);
    int st = I.osqp.solve(qp);
    if (st != -10) { std::memcpy(x, I.osqp.x.data(), qp.n * 8); std::memcpy(y, I.osqp.y.data(), qp.m * 8); }
    info6[0] = I.osqp.last_iters; info6[1] = st; info6[2] = I.osqp.last_pri; info6[3] = I.osqp.last_dua; info6[4] = I.osqp.rho; info6[5] = I.osqp.n_refactor;
    return st;
}

// ---- whole step for a batch (the five reference calls per instance): CPU baseline + closed-loop harness ---------
// solver: 0 = exact (IPM), 1 = OSQP port.  Persistent per-instance state: solved flag, previous solution, previous ts, OSQP warm start.
// outputs: u_out [B][3] (delta, Fxf, Fxr); sol_out (optional) [B][8*(N+1)] = q (6x(N+1)) then NORMALISED u (2x(N+1)); iters/status [B].
// returns wall seconds spent inside.
double po_step_batch(void* hv, int B, const double* states6, const double* controls3, const double* t0, const double* others4, const double* time_offsets,
                     int solver, int nthreads, double* u_out, double* sol_out, int* iters, int* status) {
    Handle* h = (Handle*)hv; po_reserve_instances(hv, B);
    const int Ns = h->mpc.TS.N_short, Nl = h->mpc.TS.N_long, Nn = Ns + Nl + 1;
    auto t_start = std::chrono::steady_clock::now();
    auto work = [&](int b0, int b1) {
        CoupledMPC m = h->mpc;                      // thread-local copy (TS scratch)
        LDLNumeric ldl; ldl.init(&h->kkt.sym);
        StageData sd; QP qp; Nodes nd; ExactResult R;
        for (int b = b0; b < b1; b++) {
            Inst& I = *h->inst[b];
            if (I.ts.empty()) { I.ts = m.TS.ts; I.dt = m.TS.dt; I.prev_ts = m.TS.prev_ts; }
            m.TS.ts = I.ts; m.TS.dt = I.dt; m.TS.prev_ts = I.prev_ts;
            m.TS.compute(t0[b]);                                                           // compute_time_steps!
            I.ts = m.TS.ts; I.dt = m.TS.dt; I.prev_ts = m.TS.prev_ts;
            m.linearization_nodes(states6 + 6 * b, controls3 + 3 * b, time_offsets ? time_offsets[b] : NAN, I.solved,
                                  I.prev_q.data(), I.prev_u.data(), nd);                  // compute_linearization_nodes!
            double zero4[4] = {0, 0, 0, 0};
            m.update_qp(nd, states6 + 6 * b, controls3 + 3 * b, others4 ? others4 + 4 * b : zero4, sd);   // update_QP!
            h->lay.fill(sd, m.cp, m.veh, m.u_norm, qp);
            const double* x = nullptr; int it = 0, st = 0;
            if (solver == 0) { st = solve_exact(qp, h->kkt, ldl, R); x = R.x.data(); it = R.iters; }
            else { st = I.osqp.solve(qp); x = I.osqp.x.data(); it = I.osqp.last_iters; }   // solve!
            I.prev_q.assign(x + h->lay.o_q, x + h->lay.o_q + 6 * Nn);
            I.prev_u.assign(x + h->lay.o_u, x + h->lay.o_u + 2 * Nn);
            I.solved = true;                                                               // model_predictive_control.jl:76
            m.next_control(x + h->lay.vu(0, 1), u_out + 3 * b);                            // get_next_control
            if (sol_out) { std::memcpy(sol_out + (size_t)8 * Nn * b, I.prev_q.data(), 6 * Nn * 8); std::memcpy(sol_out + (size_t)8 * Nn * b + 6 * Nn, I.prev_u.data(), 2 * Nn * 8); }
            if (iters) iters[b] = it;
            if (status) status[b] = st;
        }
    };
    if (nthreads <= 1) work(0, B);
    else {
        std::vector<std::thread> th; int per = (B + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; t++) { int b0 = t * per, b1 = std::min(B, b0 + per); if (b0 < b1) th.emplace_back(work, b0, b1); }
        for (auto& t : th) t.join();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
}

}  // extern "C"

extern "C" void po_set_hji_placeholder(void* hv) { ((Handle*)hv)->mpc.hji.placeholder(); }
extern "C" void po_set_alias_prev_ts(void* hv, int on) { ((Handle*)hv)->mpc.TS.alias_prev_ts = on != 0; }

// ---- decoupled (lateral) formulation: decoupled_lat_long.jl ------------------------------------------------------------------
namespace {
struct HandleDec { DecoupledMPC mpc; DecoupledQPLayout lay; KKTPattern kkt; int sd_len() const { return 45 * mpc.N() + 5; } };
void sdd_to_flat(const StageDataDec& sd, double* f) {
    double* p = f;
    auto cp = [&](const std::vector<double>& v) { std::memcpy(p, v.data(), v.size() * 8); p += v.size(); };
    cp(sd.A); cp(sd.B0); cp(sd.Bf); cp(sd.c); cp(sd.H); cp(sd.G); cp(sd.dmin); cp(sd.dmax); cp(sd.ddmin); cp(sd.ddmax); cp(sd.dt);
    std::memcpy(p, sd.q_curr, 32); p += 4; *p++ = sd.d_curr;
}
void flat_to_sdd(const double* f, int Ns, int Nl, StageDataDec& sd) {
    sd.resize(Ns, Nl); const double* p = f;
    auto cp = [&](std::vector<double>& v) { std::memcpy(v.data(), p, v.size() * 8); p += v.size(); };
    cp(sd.A); cp(sd.B0); cp(sd.Bf); cp(sd.c); cp(sd.H); cp(sd.G); cp(sd.dmin); cp(sd.dmax); cp(sd.ddmin); cp(sd.ddmax); cp(sd.dt);
    std::memcpy(sd.q_curr, p, 32); p += 4; sd.d_curr = *p++;
}
}  // namespace
extern "C" {
void* pd_create(int Ns, int Nl, double dt_short, double dt_long, int corr) {
    HandleDec* h = new HandleDec(); h->mpc.init(Ns, Nl, dt_short, dt_long, corr != 0);
    h->lay.build(Ns, Nl); h->kkt.build(h->lay.n, h->lay.m, h->lay.Ap, h->lay.Ai);
    return h;
}
void pd_destroy(void* hv) { delete (HandleDec*)hv; }
void pd_get_control_params(void* hv, double* v11) {
    const DecoupledControlParams& c = ((HandleDec*)hv)->mpc.cp;
    double v[11] = {c.V_min, c.V_max, c.k_V, c.k_s, c.deltadot_max, c.Q_dpsi, c.Q_e, c.W_beta, c.W_r, c.R_delta, c.R_ddelta};
    std::memcpy(v11, v, sizeof(v));
}
void pd_set_trajectory(void* hv, int L, const double* a) {
    TrajectoryTube& T = ((HandleDec*)hv)->mpc.traj; T.L = L;
    std::vector<double>* f[12] = {&T.t, &T.s, &T.V, &T.A, &T.E, &T.N, &T.psi, &T.kappa, &T.theta, &T.phi, &T.edge_L, &T.edge_R};
    for (int k = 0; k < 12; k++) f[k]->assign(a + (size_t)k * L, a + (size_t)(k + 1) * L);
}
void pd_qp_dims(void* hv, int* n, int* m, int* nnz, int* sd_len) { HandleDec* h = (HandleDec*)hv; *n = h->lay.n; *m = h->lay.m; *nnz = (int)h->lay.Ai.size(); *sd_len = h->sd_len(); }
void pd_time_steps(void* hv, double t0, double* ts, double* dt) {
    MPCTimeSteps T = ((HandleDec*)hv)->mpc.TS; T.compute(t0);
    std::memcpy(ts, T.ts.data(), T.ts.size() * 8); std::memcpy(dt, T.dt.data(), T.dt.size() * 8);
}
void pd_set_time_grid_naive(void* hv, int naive) { ((HandleDec*)hv)->mpc.TS.naive_time_grid = naive != 0; }
void pd_nodes(void* hv, const double* state6, const double* control3, double time_offset, const double* ts, const double* dt, double* qs, double* us, double* ps) {
    DecoupledMPC m = ((HandleDec*)hv)->mpc; int Nn = m.N() + 1;
    m.TS.ts.assign(ts, ts + Nn); m.TS.dt.assign(dt, dt + Nn - 1);
    NodesDec nd; m.linearization_nodes(state6, control3, time_offset, nd);
    std::memcpy(qs, nd.qs.data(), nd.qs.size() * 8); std::memcpy(us, nd.us.data(), nd.us.size() * 8); std::memcpy(ps, nd.ps.data(), nd.ps.size() * 8);
}
// (edge_L, edge_R) of the tube at every linearization node: input of the build-defined wall rows (the reference snapshot has no constraint that reads them)
void pd_node_edges(void* hv, const double* state6, const double* control3, double time_offset, const double* ts, const double* dt, double* edges) {
    DecoupledMPC m = ((HandleDec*)hv)->mpc; int Nn = m.N() + 1;
    m.TS.ts.assign(ts, ts + Nn); m.TS.dt.assign(dt, dt + Nn - 1);
    NodesDec nd; m.linearization_nodes(state6, control3, time_offset, nd);
    std::memcpy(edges, nd.edges.data(), nd.edges.size() * 8);
}
// exact solve of an arbitrary QP in the canonical form  min 1/2 x'diag(Pd)x + q'x  s.t.  l <= A x <= u  (A in CSC; rows are equalities or one-sided)
int po_solve_exact_generic(int n, int m, const double* Pd, const double* q, const int* Ap, const int* Ai, const double* Ax, const double* l, const double* u,
                           double* x, double* y, double* info5) {
    QP qp; qp.n = n; qp.m = m; qp.Pd.assign(Pd, Pd + n); qp.q.assign(q, q + n); qp.Ap.assign(Ap, Ap + n + 1); qp.Ai.assign(Ai, Ai + Ap[n]); qp.Ax.assign(Ax, Ax + Ap[n]);
    qp.l.assign(l, l + m); qp.u.assign(u, u + m);
    KKTPattern K; K.build(n, m, qp.Ap, qp.Ai);
    LDLNumeric ldl; ldl.init(&K.sym);
    ExactResult R; int st = solve_exact_robust(qp, K, ldl, R);
    if ((int)R.x.size() == n) { std::memcpy(x, R.x.data(), n * 8); std::memcpy(y, R.y.data(), m * 8); }
    info5[0] = R.iters; info5[1] = R.status; info5[2] = R.res_pri; info5[3] = R.res_dua; info5[4] = R.gap; info5[5] = R.polished;
    return st;
}
// Verification of a candidate answer of a canonical QP (any source): see polish_from in qp.hpp.  info5 as po_solve_exact_generic; status 1 only for a verified KKT point.
int po_polish_generic(int n, int m, const double* Pd, const double* q, const int* Ap, const int* Ai, const double* Ax, const double* l, const double* u,
                      const double* x0, const int* actv0, double* x, double* y, double* info5) {
    QP qp; qp.n = n; qp.m = m; qp.Pd.assign(Pd, Pd + n); qp.q.assign(q, q + n); qp.Ap.assign(Ap, Ap + n + 1); qp.Ai.assign(Ai, Ai + Ap[n]); qp.Ax.assign(Ax, Ax + Ap[n]);
    qp.l.assign(l, l + m); qp.u.assign(u, u + m);
    KKTPattern K; K.build(n, m, qp.Ap, qp.Ai);
    LDLNumeric ldl; ldl.init(&K.sym);
    ExactResult R; int st = polish_from(qp, ldl, x0, actv0, R);
    if ((int)R.x.size() == n) { std::memcpy(x, R.x.data(), n * 8); std::memcpy(y, R.y.data(), m * 8); }
    info5[0] = R.iters; info5[1] = st; info5[2] = R.res_pri; info5[3] = R.res_dua; info5[4] = R.gap; info5[5] = R.polished;
    return st;
}
void pd_update_qp(void* hv, const double* qs, const double* us, const double* ps, const double* dt, double* sd_flat) {
    DecoupledMPC m = ((HandleDec*)hv)->mpc; int Nn = m.N() + 1;
    m.TS.dt.assign(dt, dt + Nn - 1);
    NodesDec nd; nd.qs.assign(qs, qs + 4 * Nn); nd.us.assign(us, us + 2 * Nn); nd.ps.assign(ps, ps + 4 * Nn);
    StageDataDec sd; m.update_qp(nd, sd); sdd_to_flat(sd, sd_flat);
}
void pd_assemble_qp(void* hv, const double* sd_flat, double* Pd, double* q, int* Ap, int* Ai, double* Ax, double* l, double* u) {
    HandleDec* h = (HandleDec*)hv; StageDataDec sd; flat_to_sdd(sd_flat, h->mpc.TS.N_short, h->mpc.TS.N_long, sd);
    QP qp; h->lay.fill(sd, h->mpc.cp, qp);
    std::memcpy(Pd, qp.Pd.data(), qp.n * 8); std::memcpy(q, qp.q.data(), qp.n * 8);
    std::memcpy(Ap, qp.Ap.data(), (qp.n + 1) * 4); std::memcpy(Ai, qp.Ai.data(), qp.Ai.size() * 4); std::memcpy(Ax, qp.Ax.data(), qp.Ax.size() * 8);
    std::memcpy(l, qp.l.data(), qp.m * 8); std::memcpy(u, qp.u.data(), qp.m * 8);
}
int pd_solve_exact(void* hv, const double* sd_flat, double* x, double* y, double* info5) {
    HandleDec* h = (HandleDec*)hv; StageDataDec sd; flat_to_sdd(sd_flat, h->mpc.TS.N_short, h->mpc.TS.N_long, sd);
    QP qp; h->lay.fill(sd, h->mpc.cp, qp);
    LDLNumeric ldl; ldl.init(&h->kkt.sym);
    ExactResult R; int st = solve_exact_robust(qp, h->kkt, ldl, R);
    if ((int)R.x.size() == qp.n) { std::memcpy(x, R.x.data(), qp.n * 8); std::memcpy(y, R.y.data(), qp.m * 8); }
    info5[0] = R.iters; info5[1] = R.status; info5[2] = R.res_pri; info5[3] = R.res_dua; info5[4] = R.gap; info5[5] = R.polished;
    return st;
}
// whole decoupled step for a batch with the reference ALGORITHM (OSQP port, default settings, cold: the lateral formulation has no warm branch):
// time grid -> nodes (decoupled_lat_long.jl:52-104) -> update_QP! (:228-273) -> solve! -> get_next_control (:275-278).  CPU baseline of BASELINE config 5.
double pd_step_batch(void* hv, int B, const double* states6, const double* controls3, const double* t0, const double* time_offsets, int nthreads,
                     double* u_out, int* iters, int* status) {
    HandleDec* h = (HandleDec*)hv;
    auto t_start = std::chrono::steady_clock::now();
    auto work = [&](int b0, int b1) {
        DecoupledMPC m = h->mpc; const int Nn = m.N() + 1;
        OSQPSettings st_; OSQPPort osqp; osqp.init(&h->kkt, st_);
        NodesDec nd; StageDataDec sd; QP qp;
        for (int b = b0; b < b1; b++) {
            m.TS.compute(t0[b]);
            m.linearization_nodes(states6 + 6 * b, controls3 + 3 * b, time_offsets ? time_offsets[b] : NAN, nd);
            m.update_qp(nd, sd);
            h->lay.fill(sd, m.cp, qp);
            osqp.reset();
            int st = osqp.solve(qp);
            const double delta = osqp.x[4 * Nn + 1];                      // delta of node 2 (variable order: q (4 x N+1), delta (N+1), ...)
            m.next_control(delta, nd.us[2 * 1 + 1], u_out + 3 * b);
            if (iters) iters[b] = osqp.last_iters;
            if (status) status[b] = st;
        }
    };
    if (nthreads <= 1) work(0, B);
    else {
        std::vector<std::thread> th; int per = (B + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; t++) { int b0 = t * per, b1 = std::min(B, b0 + per); if (b0 < b1) th.emplace_back(work, b0, b1); }
        for (auto& t : th) t.join();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
}
void pd_lateral_dynamics(void* hv, const double* q4, const double* u2, const double* p4, double* out4) { vehicle_lateral_dynamics<double>(((HandleDec*)hv)->mpc.veh, q4, u2, p4, out4); }
void pd_linearize_interval(void* hv, const double* q4, const double* w0, const double* wf, double dt, int ramp, double* A16, double* B0, double* Bf, double* c) {
    ((HandleDec*)hv)->mpc.linearize_interval(q4, w0, wf, dt, ramp != 0, A16, B0, Bf, c);
}
void pd_next_control(void* hv, double delta, double Fx_seed, double* out3) { ((HandleDec*)hv)->mpc.next_control(delta, Fx_seed, out3); }
}
