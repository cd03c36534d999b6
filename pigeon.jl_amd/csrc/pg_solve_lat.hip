// k_solve_lat: solve! + get_next_control of the LATERAL tracking QP (decoupled_lat_long.jl:134-226,275-278) in its own 5-state stage form -- the dedicated kernel
// of BASELINE configs[4] (B = 4096, N = 50, + the build-defined wall rows).  Included by pg_kernels.hip inside namespace pg.
//
// Stage form (exact; oracle/lat_ipm_numpy.py is the numpy twin of this file's interior point):  x_k = (Uy, r, dpsi, e, delta)_k, v_k = delta_{k+1} - delta_k (the reference's d-delta
// variables, :146),  x_{k+1} = Abar_k x_k + Bbar_k v_k + cbar_k  with  Abar = [A  B0+Bf; 0 1], Bbar = [Bf; 1], cbar = [c; 0];  x_0 = (q_curr, delta_curr) fixed (:150-151).
// Rows of transition k (node k+1, input v_k), slack t >= 0 (local numbering; the bit of the 16-bit active mask pg_get_solve_info reports in brackets):
//   0 [3]: dmax - delta    1 [4]: delta - dmin    2..5 [6..9]: G_i - H_i (Uy, r) + sigma_{1,1,2,2}    6 [10]: sigma_1    7 [11]: sigma_2    8 [12]: ddmax - v    9 [13]: v - ddmin
//   walls: 10 [0]: edge_L - e + sw    11 [1]: e - edge_R + sw    12 [2]: sw
// Same interior point as k_solve (Mehrotra predictor-corrector, every Newton step ONE equality-constrained LQ problem solved by a Riccati recursion: matrix pass once per
// iteration, vector pass for the corrector, dynamics and slacks exact at every Newton point), but mapped for a 5 x 5 stage instead of being embedded in the 8 x 8 one:
//
//  * SIXTEEN lanes = one instance, four instances per wavefront (one DPP row each): 4096 instances = 1024 wavefronts = one per SIMD of the chip, a single round.
//    Round 6: or the WHOLE wavefront = one instance (template argument LPI = 64: lane = stage in the stage-parallel passes, the serial passes in the first DPP row) -- for cold
//    batches of up to 1024 instances, for the list a warm step's attempts leave, and for the stragglers a cold launch of a large batch hands over (HAND = 1 -> HAND = 2, below).
//  * Stage-parallel work (barrier terms, slack elimination, Newton point, step rules): lane c of a row owns stages c, c + 16, c + 32, c + 48 (one slot visit each per pass); t,
//    lambda and the second-order term of its rows live in a per-wavefront workspace in global memory (template argument MEM: every horizon beyond 16 intervals; L2 / Infinity
//    Cache resident, 6 TB/s through the memory side at B = 4096) -- or, with one instance per wavefront, in the lane's registers for the whole solve.
//  * Serial passes: lane c holds COLUMN c of the stage matrices ([Abar | Bbar | cbar] = columns 0..6, P = columns 0..4, the vector recursion rides in column 6), loaded
//    from the packed stage records in global memory (L2) one stage ahead (from an LDS copy where the wavefront has one instance and the room).  Every product is a sequence
//    of v_fmac_f64_dpp row_newbcast:k -- the broadcast of lane k's register to its row fused into the multiply-add (tools/probes/dpp_f64_probe.hip: 8.5 cycles against 22.5
//    for v_mov_b64_dpp + v_fma_f64, which is what the compiler emits for the builtin) -- no LDS traffic and no cross-lane shuffles in the dependent chain.  A stage of the
//    matrix pass is 181 instructions (79 of them DPP multiply-adds; round 6 took it from 193: three operand sets in rotation instead of register copies, per-lane strides as
//    lane masks, store addresses chosen once -- EXPERIMENTS 12.10), 1000-1100 cycles with one wavefront per SIMD.
//    P is used from both orientations in M = P [Abar Bbar cbar] (M = (P + P')/2 X): the antisymmetric rounding error of the recursion never propagates (k_solve
//    symmetrises through LDS every stage at N = 50).
//  * The roll-out holds ROW i of [Abar | Bbar | cbar] in lane i, the delta row in lane 4 and the gain row in lane 5 -- every lane forms its row as (matrix side) + (LDS
//    side) through the same two loads with its own base and stride -- : x_{k+1} = six fused multiply-adds per stage, 47 instructions.
//  * LDS per instance: 24 doubles per stage (stage cost terms that the roll-out result overwrites, gains, S^-1, P cbar) = 9.6 KB at N = 50: four wavefronts per CU.
//  * STRAGGLER HAND-OVER (round 6, at the kernel itself): a cold batch needs 13 trips through the loop on average and 26-29 for its slowest instance.  The first launch
//    (HAND = 1) stops after a fixed number of trips and files its unfinished instances; the second (HAND = 2) resumes them one per wavefront (2.96-3.03 -> 2.47-2.50 ms).
//
//  * Round 5 -- PINNED INPUTS.  The stage has ONE input, and a held steering-rate row fixes it (v = +ddmax / -ddmin).  The polish eliminates such a row exactly instead of
//    iterating on its multiplier: the stage cost of a pinned stage carries Rhat = BIGP, rhat = -BIGP v -- a penalty so large that 1 / BIGP vanishes against everything else
//    in the arithmetic (1e200; 1e22 in fp32) -- and the UNCHANGED serial passes return K = 0, kff = v to the last bit.  The row's multiplier is read off stationarity in v by
//    the stage's own lane behind the roll-out: lambda = -/+ (F x + Bbar'y + S v), with F = Bbar'P Abar, Bbar'P Bbar and Bbar'y left in a small per-wavefront block of the
//    workspace by the matrix (vector) pass.  Round 4's augmented Lagrangian contracted by S / (S + rho) per pass with S up to 1e12 on the open-loop unstable 8 s horizon, and such
//    multipliers stalled: 53 / 54 of 4096 config-5 answers ended unverified; now 0 / 4.
//    (First version of this round: K = 0 / kff = v through flags in every serial pass, each pass compiled twice -- +3.5 % on the launch.  Measured and removed: pinning a held
//    steering-BOUND row the same way, v = -delta_k + bound, i.e. the fixed feedback K = -e_4' with the general recursion P_k = Q + A'PA + F'K + K'F + K'SK -- the same 4095 /
//    4096 verify, but 24 answers move by 1e-7 .. 8e-7 and delta by up to 6e-4 in the far horizon: those rows stay penalised.)
//
// The DPP forms are inline assembly: hipcc does not pad their hazards (VALU write -> DPP read of the same register: 2 wait states; EXEC write -> DPP: 5), so every block opens
// with s_nop 4 and the pass loops carry no divergent branch (predicated stores go to a dummy LDS slot).

#ifdef PG_F32
#define LAT_FMAC "v_fmac_f32_dpp"
#define LAT_MOV "v_mov_b32"
#else
#define LAT_FMAC "v_fmac_f64_dpp"
#define LAT_MOV "v_mov_b64"
#endif
#define LAT_DPP(acc, bsrc, x, lane) LAT_FMAC " %" #acc ", %" #bsrc ", %" #x " row_newbcast:" #lane " row_mask:0xf bank_mask:0xf\n\t"

#ifndef LAT_SYM
#define LAT_SYM 1                 // the matrix pass multiplies by 1/2 (P + P'): experiment switch
#endif
#ifndef LAT_PREFETCH_V
#define LAT_PREFETCH_V 3        // stages of matrix columns in flight in the vector pass
#endif
#ifndef LAT_PREFETCH_F
#define LAT_PREFETCH_F 3        // stages of matrix rows in flight in the roll-out
#endif
constexpr int LAT_REC = 12;     // stage -> serial: (yy, yr, rr, ee, dd, Rhat) [overwritten by the roll-out: x+_{k+1}[0..4], v+_k], qhat[5], rhat
constexpr int LAT_TAB = 12;     // serial tables: K[5], kff, Sinv, P cbar [5]
constexpr int LAT_STRIDE = LAT_REC + LAT_TAB;
constexpr int LAT_XS = 34;      // one instance per wavefront: the stage matrices (32 doubles: rows of [A | B0+Bf | Bf | c]) and dt of a stage are copied into the LDS once per launch, 34 doubles apart
__host__ __device__ inline size_t lat_lds_doubles(int N, int instances_per_wavefront = 4) {
    return (size_t)instances_per_wavefront * N * LAT_STRIDE + 72 + 16 + (instances_per_wavefront == 1 ? (size_t)N * LAT_XS : 0); }
// horizons beyond 32 intervals keep the per-row interior-point state in a global workspace (see "stage-parallel part"): bytes per 16-stage slot of one wavefront
// (13 rows x 64 lanes x (t, lambda) + 7 x 64 x 2 second-order terms + 64 x 4 eliminated slacks + 64 x 16 B of working-set words, fp64), four slots per wavefront
constexpr size_t LAT_WS_SLOT_BYTES = 23552;
__host__ __device__ inline size_t lat_ws_bytes(int B) { return (size_t)((B + 3) / 4) * 4 * LAT_WS_SLOT_BYTES; }
// what the multiplier of a pinned rate row is read from (C.lat_aux, per instance and stage): F[0..4] = Bbar'P Abar, Bbar'P Bbar, Bbar'y, -   (L2-resident, written by every matrix pass)
constexpr int LAT_AUX = 8;
// Stage constants of the STAGE-PARALLEL passes in the order the wavefront reads them (round 6): C.lat_spc [home wavefront][slot][LAT_SPC_Q][64 lanes] real2 -- H (4), G (2), the
// steering bounds, the rate bounds, (dt, 0) of the stage a lane owns in that slot.  Every slot visit of every pass used to read these 136 B per lane from the packed records,
// lane = stage: each of the nine load instructions touched 64-128 cache lines, twelve visits per trip, four wavefronts per CU -- the address unit, not the bytes, was what the
// visits queued for.  The wavefront that starts an instance files the constants once, lane-contiguous: the same nine loads now touch eight lines each.
constexpr int LAT_SPC_Q = 9;
__host__ __device__ inline size_t lat_spc_bytes(int B, int N) { return (size_t)((B + 3) / 4) * ((N + 15) / 16) * LAT_SPC_Q * 64 * 2 * sizeof(real); }

// broadcast of lane K of each 16-lane row (compiler builtin: hazards padded by hipcc; v_mov_b64_dpp row_newbcast)
template <int K> PG_DEV real lat_bc(real v) {
#ifdef PG_F32
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xF, 0xF, false));
#else
    return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + K, 0xF, 0xF, false));
#endif
}
// all-reduce over the 16 lanes of a row (DPP butterflies: no LDS)
PG_DEV real row_sum(real v) { v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x124>(v); v += dpp_move<0x128>(v); return v; }
PG_DEV real row_max(real v) { v = fmax(v, dpp_move<0xB1>(v)); v = fmax(v, dpp_move<0x4E>(v)); v = fmax(v, dpp_move<0x124>(v)); v = fmax(v, dpp_move<0x128>(v)); return v; }
// all-reduce over the LPI lanes that serve one instance in the stage-parallel passes: a DPP row (16), or the whole wavefront (64: one instance per wavefront, round 6)
// (round 6, tried and dropped: forming the sums of the two arrangements in ONE order -- per-slot subtotals, lanes l % 16, + 16, + 32, + 48 first, then the row butterfly -- so that
//  an instance takes the same iterations to the same bits in either.  The sums then agree; the per-row arithmetic does not: the compiler contracts multiply-adds differently in the
//  rolled slot loop of the workspace variant and in the single visit of the register variant -- multipliers one ulp apart after the FIRST iteration on all 256 instances of a
//  test batch (tools/gpu_lat_arrangements.py).  The arrangements agree to 1e-7 in the answer, not bit for bit; which arrangement runs is a rule of the data alone.)
template <int LPI> PG_DEV real grp_sum(real v) { v = row_sum(v); if constexpr (LPI == 64) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); } return v; }
template <int LPI> PG_DEV real grp_max(real v) { v = row_max(v); if constexpr (LPI == 64) { v = fmax(v, __shfl_xor(v, 16)); v = fmax(v, __shfl_xor(v, 32)); } return v; }
// reciprocal of the interior-point weights: hardware seed + ONE Newton step (~1e-14 relative in fp64; the Newton system only has to be consistent, see assemble)
PG_DEV real lat_rcp(real x) {
#ifdef PG_F32
    return frcp(x);
#else
    double r = __builtin_amdgcn_rcp(x); double e = fma(-x, r, 1.0); return fma(r, e, r);
#endif
}

// packed stage records from the embedded QP block (pg_set_qp installs QP data behind k_qp_dec's back): thread = (instance, stage)
__global__ __launch_bounds__(128) void k_lat_pack(DevCfg C, int b0, int n, const real* __restrict__ qp) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)n * C.N) return;
    const int b = b0 + (int)(gid / C.N), t = (int)(gid % C.N);
    const QpOff o = qp_offsets(C.N);
    const real* Q = qp + (size_t)b * C.qp_len;
    real* Lp = C.lat_pack + ((size_t)b * C.N + t) * LATP;
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) Lp[8 * i + j] = Q[o.A + 36 * t + 6 * (2 + i) + 2 + j];
        const real b0v = Q[o.B0 + 12 * t + 2 * (2 + i)], bfv = Q[o.Bf + 12 * t + 2 * (2 + i)];
        Lp[8 * i + 4] = b0v + bfv; Lp[8 * i + 5] = bfv; Lp[8 * i + 6] = Q[o.c + 6 * t + 2 + i]; Lp[8 * i + 7] = real(0.0);
        Lp[32 + 2 * i] = Q[o.H + 8 * t + 2 * i]; Lp[33 + 2 * i] = Q[o.H + 8 * t + 2 * i + 1]; Lp[40 + i] = Q[o.G + 4 * t + i];
    }
    Lp[44] = Q[o.dmax + t]; Lp[45] = Q[o.dmin + t]; Lp[46] = Q[o.ddmax + t]; Lp[47] = Q[o.ddmin + t]; Lp[48] = Q[o.dt + t];
    for (int i = 49; i < LATP; i++) Lp[i] = real(0.0);
}

// Round 6 -- STRAGGLER HAND-OVER (SolveOut::hand_mode).  A cold N = 50 batch needs 14 trips through the loop on average and 29 for its slowest instance, four instances
// share a wavefront, and every wavefront is resident from the first cycle: after trip 17 a quarter of the instances but two thirds of the wavefronts are still alive, each at
// the full price of a trip.  hand_mode = 1: a wavefront stops at a trip boundary once the batch has few enough unfinished instances (hand_target, counted on the device;
// or after hand_cap trips), files the scalars of its unfinished instances in a small record (their row state already lives in the workspace) and appends them to the to-do
// list.  hand_mode = 2: the launch behind it RESUMES the listed instances -- with LPI = 64 ONE instance per wavefront: the stage-parallel passes see one stage per lane (one
// slot visit instead of four, the row state in registers), the serial passes run as before in the first DPP row.  A trip then costs a wavefront about half of what it costs
// four instances sharing one, and the tail of the batch runs on a few hundred wavefronts instead of two thirds of the chip.
constexpr int LAT_HAND_R = 8, LAT_HAND_I = 16;      // per-instance hand-over record: reals (mu, phi, rp0, mu0i, tol_cur, tol_cold), ints (see the save below)
template <int NSLOT, bool WALLS, bool MEM, int LPI = 16, int HAND = 0>
__global__ __launch_bounds__(64, 1) void k_solve_lat(DevCfg C, int B, const real* __restrict__ qp, const real* __restrict__ nodes, SolveOut O, unsigned long long* __restrict__ prof) {
    static_assert(LPI == 16 || (LPI == 64 && NSLOT == 1 && !MEM), "one instance per wavefront keeps its single slot in registers");
    static_assert(NSLOT == 1, "the two-slot register instantiation (horizons of 17..32 intervals) is retired: hipcc 7.2 allocated one accumulation register to two live values in it (see configure_lateral)");
    static_assert(HAND == 0 || (HAND == 1 && MEM && LPI == 16) || (HAND == 2 && LPI == 64), "hand-over: out of a launch whose row state lives in the workspace, into one instance per wavefront");
    // (LPI = 64 with HAND = 0: a small batch, one instance per wavefront from the start)
    // (HAND is a template argument, not a launch argument: the plain kernel carries none of the hand-over's code -- as a run-time flag it cost the round-5 launch 9 %)
    constexpr int NR = WALLS ? 13 : 10;
    constexpr int NI = 64 / LPI;                          // instances per wavefront
    // diagnostic cycle counters (pg_debug_solve_cycles): 0 barrier terms, 1 matrix pass, 2 vector pass, 3 roll-outs, 4 Newton point / step rules, 5 everything else
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0}, tprev = prof ? clock64() : 0;
#ifdef LAT_MP_TIMING
    unsigned long long mp[4] = {0, 0, 0, 0};      // experiment: sub-phases of a matrix-pass stage (requests, M, G + reciprocal + K, P)
#define LAT_MPT(i) do { if (prof) { const unsigned long long n_ = clock64(); mp[i] += n_ - mpt; mpt = n_; } } while (0)
#else
#define LAT_MPT(i) do { } while (0)
#endif
    auto stamp = [&](int slot) __attribute__((always_inline)) { if (prof) { const unsigned long long now = clock64(); pc[slot] += now - tprev; tprev = now; } };
    // c: the lane's role in the serial passes (column / row of the stage matrices inside its DPP row); cs: its role in the stage-parallel passes (stage cs, cs + LPI, ...);
    // frow: the lanes that run the serial passes (LPI = 64: ONE DPP row serves the wavefront's one instance; the other three rows sit the pass out under EXEC -- their loads
    // would cost the CU's address unit what the first row's do)
    const int lane = threadIdx.x, g = LPI == 16 ? lane >> 4 : 0, c = lane & 15, cs = LPI == 16 ? c : lane;
    const bool frow = LPI == 16 || lane < 16;
    const int N = C.N, NN = C.NN;
    constexpr bool resume = HAND == 2;
    // Two launches per WARM step (round 4, SolveOut::todo / list): the first runs the warm attempts only and appends what they do not serve (and the instances that are
    // cold or backing off) to a list; the second solves that list cold, four instances per wavefront again -- 150 wavefronts instead of the 600 that would otherwise
    // stay alive for one unserved instance each, i.e. one per CU instead of two or three: a pass through the loop costs 61-80 us there against 110 us with four per CU.
    const bool listm = O.n_list != nullptr, defer = O.todo != nullptr && O.hand_mode == 0;
    // (a resumed launch, HAND = 2: the list has two ends -- the instances that were still in their interior point at the hand-over, which have the most trips ahead of them,
    //  are filed from the front and start first; the ones already in their polish, one to three trips from done, from the back of the same array, counted in n_list[2])
    const int n_front = listm ? *O.n_list : B;
    const int n_list = (listm && HAND == 2) ? n_front + O.n_list[2] : n_front;
    if (listm && NI * (int)blockIdx.x >= n_list) return;          // (uniform over the block; nothing has been touched yet)
    if (listm && HAND == 0 && (n_list < O.list_lo || (O.list_hi > 0 && n_list > O.list_hi))) return;      // (a list of this length belongs to the other arrangement's launch)
    const int idx_raw = NI * (int)blockIdx.x + g;
    const int b_raw = listm ? (idx_raw < n_list ? (HAND == 2 && idx_raw >= n_front ? O.list[B - 1 - (idx_raw - n_front)] : O.list[idx_raw]) : B) : idx_raw;
    const bool valid = b_raw < B;
    const int b = valid ? b_raw : B - 1;                 // (a ragged last wavefront solves the last instance again and stores nothing)
    extern __shared__ real lds[];
    real* const sI = lds + (size_t)g * N * LAT_STRIDE;   // this instance's region: rec[N][12] then tab[N][12]
    real* const sRec = sI; real* const sTab = sI + (size_t)N * LAT_REC;
    real* const sDum = lds + (size_t)NI * N * LAT_STRIDE; // [72] sink for predicated-off stores (lane + up to four slots)
    real* const sZero = sDum + 72;                       // a stored 0
#ifdef PG_DIAG
    if (C.dbg_poison) { for (int i = lane; i < (int)lat_lds_doubles(N, NI == 1 ? 1 : 4); i += 64) lds[i] = real(NAN); __syncthreads(); }
#endif
    if (lane < 8) sZero[lane] = real(0.0);
    constexpr bool XLDS = LPI == 64;                     // the serial passes read the stage matrices from the LDS (a wavefront with ONE instance has the room: 13.6 KB at N = 50)
    real* const sDelta = sZero + 8;                       // [8] the roll-out's delta row: (0, 0, 0, 0, 1, 1, 0, 0)
    if (lane < 8) sDelta[lane] = (lane == 4 || lane == 5) ? real(1.0) : real(0.0);
    real* const sMat = sZero + 16;
    const QpOff o = qp_offsets(N);
    const real* const Q = qp + (size_t)b * C.qp_len;

    // ---------------- addressing of the serial passes ----------------
    // The stage matrices come from the packed records k_qp_dec wrote (LATP doubles per stage, L2-resident): [8 i + m] = row i of [A | B0+Bf | Bf | c], m = 7 a stored 0.
    const real* const Lb = C.lat_pack + (size_t)b * N * LATP;
    // (an idle lane group -- the tail of a ragged last wavefront, whose lanes run instance B - 1 again and store nothing else -- gets the spare block behind the batch: in a
    //  list-mode launch the real instance B - 1 may be in ANOTHER wavefront, or long done in another launch, and a second writer of its block is a race: round 6, found by the
    //  bit-for-bit test of the closed loop with walls -- the multipliers of instance 4095's pinned rows 3e-5 apart between two runs)
    real* const aux = C.lat_aux + (size_t)(valid ? b : B) * 64 * LAT_AUX;
    // column distribution (matrix + vector pass): lane c holds X[0..3][c] of [Abar | Bbar | cbar]; row 4 is the constant x4; lanes 7..15 read the zero column
    constexpr int XSTR = XLDS ? LAT_XS : LATP, XDT = XLDS ? 32 : 48;
    const real* const Xb = XLDS ? sMat : Lb;
    const real* const colp = Xb + (c < 7 ? c : 7);
    const real cx4 = (c == 4 || c == 5) ? real(1.0) : real(0.0);
    const real cpsi = c == 2 ? real(2.0) * C.cp.Q_dpsi : real(0.0);      // Qhat[2][2] = 2 Q_dpsi dt_k is not stored: lane 2 forms it from dt_k
    // table slot lane c writes after a stage of the matrix pass: K[c] (c < 5), kff (lane 6 -> slot 5), Sinv (lane 5 -> slot 6)
    const int wslot = c < 5 ? c : (c == 6 ? 5 : (c == 5 ? 6 : -1));
    const real m6 = c == 6 ? real(1.0) : real(0.0), m5lt = c < 5 ? real(1.0) : real(0.0);

    // ---------------- Riccati matrix pass (+ the predictor's vector recursion in column 6) ----------------
    // Software-pipelined by hand: the operands of stage k - 1 (global: the matrix column and dt; LDS: the stage-cost column, Rhat / rhat) are requested at the top of
    // stage k and first touched at the top of stage k - 1, a whole stage of arithmetic (~900 cycles) later.
    auto matrix_pass = [&](auto aux_on) __attribute__((always_inline)) {      // aux_on: leave F / B'PB / B'y of every stage in lat_aux (what a pinned row's multiplier is read from)
        // (the lane's offsets and masks are formed HERE, per call, not once per kernel: kept alive across the whole solve they are parked in accumulation registers and read back at
        //  every use -- 22 reads per stage; the volatile asm keeps the compiler from hoisting them out again)
        // stage-cost column of lane c inside rec[k]: Qhat[i][c] for c < 5, qhat[i] for c == 6, else zero (a stored 0 with stride 0)
        // (a per-lane stride is either 0 or one constant: `uniform product & lane mask` -- one v_and with a scalar operand -- instead of a 32-bit multiplication per lane and
        //  stage, which this hardware issues at a quarter of the rate: round 6)
        int qoff[5], qmul[5];      // (qmul: the lane MASK, -1 where the lane's column entry is stored, 0 where it reads the stored zero)
    #pragma unroll
        for (int i = 0; i < 5; i++) {
            int e = -1;
            if (c == 0) e = i == 0 ? 0 : (i == 1 ? 1 : -1);
            else if (c == 1) e = i == 0 ? 1 : (i == 1 ? 2 : -1);
            else if (c == 3) e = i == 3 ? 3 : -1;
            else if (c == 4) e = i == 4 ? 4 : -1;
            else if (c == 6) e = 6 + i;
            qoff[i] = e >= 0 ? (int)(sRec - sZero) + e : 0; qmul[i] = e >= 0 ? -1 : 0;
            asm volatile("" : "+v"(qmul[i]));      // (opaque: the compiler turns a known 0 / -1 mask back into a condition held in scalar registers -- and spills those)
        }
        const int roff = c == 5 ? (int)(sRec - sZero) + 5 : (c == 6 ? (int)(sRec - sZero) + 11 : 0), rmul_ = (c == 5 || c == 6) ? -1 : 0;      // Rhat (lane 5) / rhat (lane 6)
        int rmul = rmul_; asm volatile("" : "+v"(rmul));
        // (predicated-off stores go to the lane's sink slot: an offset and a lane mask chosen once, not a pointer select per store)
        const int mstoff = c == 6 ? (int)(sTab - sZero) + 7 : (int)(sDum - sZero) + lane, wstoff = wslot >= 0 ? (int)(sTab - sZero) + wslot : (int)(sDum - sZero) + lane;
        int mstmsk = c == 6 ? -1 : 0, wstmsk = wslot >= 0 ? -1 : 0; asm volatile("" : "+v"(mstmsk), "+v"(wstmsk));

        // Three operand sets in rotation (round 6): a stage reads P from the Q slots of set `a` (where the stage before accumulated it), computes on the operands of set `b`
        // -- whose Q slots it turns into the next P in place -- and requests the operands of the stage after into set `c`.  Rolled with two sets and P / Pn, the loop
        // copied 23 doubles per stage from the "next" names to the "current" ones (v_mov_b64: a ninth of its instructions); written out three stages per trip, nothing moves.
        struct MpSet { real X[4], Q[5], dt, r; };
        MpSet s0, s1, s2;
        {   // P_N = Qhat_{N-1} (cost on node N), p_N = qhat_{N-1}
            const real dtl = Xb[(size_t)(N - 1) * XSTR + XDT];
#pragma unroll
            for (int i = 0; i < 5; i++) s0.Q[i] = (sZero + qoff[i])[(LAT_REC * (N - 1)) & qmul[i]];
            s0.Q[2] += cpsi * dtl;
        }
        auto request = [&](int k, MpSet& o) __attribute__((always_inline)) {      // operands of stage k (k < 0: stage 0 again, unused)
            const int kk = k < 0 ? 0 : k, km = kk > 0 ? kk - 1 : 0;
            const real* cp = colp + (size_t)kk * XSTR;
#pragma unroll
            for (int i = 0; i < 4; i++) o.X[i] = cp[8 * i];
            o.dt = Xb[(size_t)km * XSTR + XDT];                // dt of stage k - 1: its cost sits on node k
#pragma unroll
            for (int i = 0; i < 5; i++) o.Q[i] = (sZero + qoff[i])[(LAT_REC * km) & qmul[i]];
            o.r = (sZero + roff)[(LAT_REC * kk) & rmul];
        };
        auto stage = [&](int k, const real* P, MpSet& cur, MpSet& nxt) __attribute__((always_inline)) {
#ifdef LAT_MP_TIMING
            unsigned long long mpt = prof ? clock64() : 0;
#endif
            request(k - 1, nxt);
            LAT_MPT(0);
            real* const X = cur.X; real* const Pn = cur.Q;
            real Xh[5], M[5];
#pragma unroll
            for (int i = 0; i < 4; i++) { Xh[i] = real(LAT_SYM ? 0.5 : 1.0) * X[i]; M[i] = real(0.0); }
            Xh[4] = real(LAT_SYM ? 0.5 : 1.0) * cx4; M[4] = real(0.0);
            // M[i][c] = sum_k 1/2 (P[i][k] + P[k][i]) X[k][c]:  bc_k(P[i]) = P[i][k],  bc_i(P[k]) = P[k][i]
            asm volatile("s_nop 4\n\t"
#define LAT_M1(k, xk) LAT_DPP(0, 5, xk, k) LAT_DPP(1, 6, xk, k) LAT_DPP(2, 7, xk, k) LAT_DPP(3, 8, xk, k) LAT_DPP(4, 9, xk, k)
#define LAT_M2(pk, xk) LAT_DPP(0, pk, xk, 0) LAT_DPP(1, pk, xk, 1) LAT_DPP(2, pk, xk, 2) LAT_DPP(3, pk, xk, 3) LAT_DPP(4, pk, xk, 4)
#if LAT_SYM
                         LAT_M1(0, 10) LAT_M2(5, 10) LAT_M1(1, 11) LAT_M2(6, 11) LAT_M1(2, 12) LAT_M2(7, 12) LAT_M1(3, 13) LAT_M2(8, 13) LAT_M1(4, 14) LAT_M2(9, 14)
#else
                         LAT_M1(0, 10) LAT_M1(1, 11) LAT_M1(2, 12) LAT_M1(3, 13) LAT_M1(4, 14)
#endif
                         : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "+v"(M[3]), "+v"(M[4])
                         : "v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]), "v"(P[4]), "v"(Xh[0]), "v"(Xh[1]), "v"(Xh[2]), "v"(Xh[3]), "v"(Xh[4]));
            LAT_MPT(1);
            // lane 6: M = P cbar (kept for the corrector's vector pass), then y = P cbar + p
            { real* const mp_ = sZero + mstoff + ((LAT_TAB * k) & mstmsk);      // lane 6: sTab[k][7..11]; the others: their five sink slots (sDum is 72 long for this)
#pragma unroll
              for (int i = 0; i < 5; i++) { mp_[i] = M[i]; M[i] = fma(m6, P[i], M[i]); } }
            // G[c] = sum_i Bbar[i] M[i][c]: lanes 0..4 F, lane 5 Bbar' P Bbar, lane 6 Bbar' y   (Bbar = column 5; its row 4 is 1)
            real G0 = real(0.0), G1 = M[4];
            asm volatile("s_nop 4\n\t"
                         LAT_DPP(0, 2, 6, 5) LAT_DPP(1, 3, 7, 5) LAT_DPP(0, 4, 8, 5) LAT_DPP(1, 5, 9, 5)
                         : "+v"(G0), "+v"(G1)
                         : "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(X[3]), "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]));
            // (the stage-cost term is added LAST: on a pinned stage it is BIGP (x the pinned value) and would swallow Bbar'P Bbar / Bbar'y, which the row's multiplier needs)
            const real Gb = G0 + G1;                      // lanes 0..4: F; lane 5: Bbar' P Bbar; lane 6: Bbar' y
            if constexpr (decltype(aux_on)::value) aux[(size_t)k * LAT_AUX + (c < 7 ? c : 7)] = Gb;      // (no branch in a pass loop: lanes 7..15 write the unused eighth slot; a wave-uniform "some instance is in a polish" test around the store measured SLOWER, 3.096 against 3.026 ms)
            const real G = Gb + cur.r;                    // lane 5: S = Rhat + Bbar' P Bbar; lane 6: f = rhat + Bbar' y
            const real Sinv = frcp(lat_bc<5>(G));
            const real Kc = -G * Sinv;                    // lanes 0..4: K[c]; lane 6: kff
            (sZero + wstoff)[(LAT_TAB * k) & wstmsk] = c == 5 ? Sinv : Kc;
            LAT_MPT(2);
            // P_k[i][c] = Qhat[i][c] + sum_k Abar[k][i] M[k][c] + F[i] K[c]   (lane 6: p_k = qhat + Abar' y + F kff);  Abar[4][i] = (i == 4): accumulated IN the Q slots of this set
            Pn[2] += cpsi * cur.dt;
            Pn[4] += M[4];                                // row 4 of Abar is e_4'
            asm volatile("s_nop 4\n\t"
#define LAT_P1(xk, mk) LAT_DPP(0, xk, mk, 0) LAT_DPP(1, xk, mk, 1) LAT_DPP(2, xk, mk, 2) LAT_DPP(3, xk, mk, 3) LAT_DPP(4, xk, mk, 4)
                         LAT_P1(5, 9) LAT_P1(6, 10) LAT_P1(7, 11) LAT_P1(8, 12) LAT_P1(13, 14)
                         : "+v"(Pn[0]), "+v"(Pn[1]), "+v"(Pn[2]), "+v"(Pn[3]), "+v"(Pn[4])
                         : "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(X[3]), "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(G), "v"(Kc));
            LAT_MPT(3);
        };
        request(N - 1, s1);
        int k = N - 1;
#pragma unroll 1
        while (true) {
            stage(k, s0.Q, s1, s2); if (--k < 0) break;
            stage(k, s1.Q, s2, s0); if (--k < 0) break;
            stage(k, s2.Q, s0, s1); if (--k < 0) break;
        }
    };

    // ---------------- Riccati vector pass (corrector): p_k = qhat + Abar' y + K f,  y = P cbar + p,  f = rhat + Bbar' y,  kff = -Sinv f ----------------
    // A stage is ~25 instructions here: the matrix columns are requested THREE stages ahead (L2 latency ~ several stages of this pass), the LDS operands one.
    auto vector_pass = [&](auto aux_on) __attribute__((always_inline)) {
        constexpr int D = LAT_PREFETCH_V;
        real buf[D][4];
        const int mcoff = c < 5 ? (int)(sTab - sZero) + 7 + c : 0, mcmul_ = c < 5 ? -1 : 0;       // P cbar [c]
        int mcmul = mcmul_; asm volatile("" : "+v"(mcmul));
        const int koff = c < 5 ? (int)(sTab - sZero) + c : 0;                                        // K[c]
        const int qvoff = c < 5 ? (int)(sRec - sZero) + 6 + c : 0, qvmul_ = c < 5 ? -1 : 0;      // qhat[c]
        int qvmul = qvmul_; asm volatile("" : "+v"(qvmul));
        real p = (sZero + qvoff)[(LAT_REC * (N - 1)) & qvmul];
        auto request = [&](int k, real* Xo) __attribute__((always_inline)) { const real* cp = colp + (size_t)(k < 0 ? 0 : k) * XSTR;
#pragma unroll
            for (int i = 0; i < 4; i++) Xo[i] = cp[8 * i]; };
        auto request_lds = [&](int k, real* o5) __attribute__((always_inline)) { const int kk = k < 0 ? 0 : k, km = kk > 0 ? kk - 1 : 0;
            o5[0] = (sZero + mcoff)[(LAT_TAB * kk) & mcmul]; o5[1] = (sZero + koff)[(LAT_TAB * kk) & mcmul]; o5[2] = (sZero + qvoff)[(LAT_REC * km) & qvmul]; o5[3] = sRec[LAT_REC * kk + 11]; o5[4] = sTab[LAT_TAB * kk + 6]; };
#pragma unroll
        for (int u = 0; u < D; u++) request(N - 1 - u, buf[u]);
        real lo[5], ln[5];
        request_lds(N - 1, lo);
#pragma unroll 1
        for (int k0 = N - 1; k0 >= 0; k0 -= D) {
            // ONE basic block per D stages (stages below 0 of the last group run on stage 0's operands and store nothing): a buffer is re-requested only after the
            // arithmetic that reads it, so the loads land in the registers they replace and nothing has to be copied (a copy would wait for the load it copies)
#pragma unroll
            for (int u = 0; u < D; u++) {
                const int k = k0 - u;
                request_lds(k - 1, ln);
                const real y = lo[0] + p;              // lanes >= 5: 0
                real acc = cx4 * lat_bc<4>(y);         // row 4 of columns 4, 5 is 1
                asm volatile("s_nop 4\n\t"
                             LAT_DPP(0, 1, 2, 0) LAT_DPP(0, 1, 3, 1) LAT_DPP(0, 1, 4, 2) LAT_DPP(0, 1, 5, 3)
                             : "+v"(acc) : "v"(y), "v"(buf[u][0]), "v"(buf[u][1]), "v"(buf[u][2]), "v"(buf[u][3]));
                request(k - D, buf[u]);
                const real by = lat_bc<5>(acc);        // lane 5's column is Bbar: Bbar'y
                if constexpr (decltype(aux_on)::value) aux[(size_t)(k < 0 ? 0 : k) * LAT_AUX + ((c == 0 && k >= 0) ? 6 : 7)] = by;
                const real f = lo[3] + by;
                *((c == 0 && k >= 0) ? sTab + LAT_TAB * (k < 0 ? 0 : k) + 5 : sDum + lane) = -lo[4] * f;
                p = m5lt * (lo[2] + acc + lo[1] * f);
#pragma unroll
                for (int i = 0; i < 5; i++) lo[i] = ln[i];
            }
        }
    };

    // ---------------- roll-out: lane i < 4 holds row i of [A | B0+Bf | Bf | c], lane 4 the delta row, lane 5 the gain row (K | 0 | kff) ----------------
    const real x0c = c < 4 ? Q[o.qcurr + 2 + c] : (c == 4 ? Q[o.ucurr] : real(0.0));
    auto forward_pass = [&](bool use_gain) __attribute__((always_inline)) {
        constexpr int D = LAT_PREFETCH_F;
        // Row of lane c = (what it reads from the matrix side) + (what it reads from the LDS side), every lane through the same two loads with ITS OWN base and stride
        // (round 6: the per-element selects "gain row or matrix row" and the multiplications by a 0 / 1 lane weight were 19 of a stage's 53 instructions):
        //   lanes 0..3: row c of [A | B0+Bf | Bf | c]  +  a stored zero row         lane 4: zero row  +  the delta row (0, 0, 0, 0, 1, 1, 0)
        //   lane 5: zero row  +  the gain row (K | . | kff) of the table, or a zero row before the first matrix pass (v = 0)        lanes 6..15: zero + zero
        // (the table keeps kff in slot 5, where the row wants a 0 and the constant term one slot later: element 5 is taken with a lane weight, element 6 through its own address)
        const bool isK = c == 5 && use_gain;
        real xr = x0c;
        // (LDS addresses are INTEGER offsets from one LDS base: a pointer chosen per lane would lose its address space -- flat loads, and they were)
        int gs = c < 4 ? -1 : 0; asm volatile("" : "+v"(gs));                                       // (lane masks, as above)
        const int goff = c < 4 ? (int)(sMat - sZero) + 8 * c : 0;                                  // (matrices in the LDS: one instance per wavefront)
        const real* const gsrc = c < 4 ? Lb + 8 * c : C.lat_zero;                                  // (matrices in the packed records: two global pointers)
        const int loff = isK ? (int)(sTab - sZero) : (c == 4 ? (int)(sDelta - sZero) : 0);
        const int loff6 = isK ? loff + 5 : loff + 6;
        int ls = isK ? -1 : 0; asm volatile("" : "+v"(ls));
        const real w5 = c == 5 ? real(0.0) : real(1.0);
        real2 buf[D][4], kt[3], ktn[3]; real k6, k6n;
        auto request = [&](int k, real2* o4) __attribute__((always_inline)) {
            const int kc = k < N ? k : N - 1;
            const real2* rp;
            if constexpr (XLDS) rp = reinterpret_cast<const real2*>(sZero + goff + ((XSTR * kc) & gs)); else rp = reinterpret_cast<const real2*>(gsrc + (unsigned)((XSTR * kc) & gs));
#pragma unroll
            for (int q = 0; q < 4; q++) o4[q] = rp[q]; };
        auto request_lds = [&](int k, real2* o3, real& o6) __attribute__((always_inline)) { const int kc = k < N ? k : N - 1; const real2* tp = reinterpret_cast<const real2*>(sZero + loff + ((LAT_TAB * kc) & ls));
#pragma unroll
            for (int q = 0; q < 3; q++) o3[q] = tp[q];
            o6 = (sZero + loff6)[(LAT_TAB * kc) & ls]; };
#pragma unroll
        for (int u = 0; u < D; u++) request(u, buf[u]);
        request_lds(0, kt, k6);
#pragma unroll 1
        for (int k0 = 0; k0 < N; k0 += D) {
            // (one basic block per D stages, as in the vector pass; stages >= N of the last group run on the last stage's operands and store nothing)
#pragma unroll
            for (int u = 0; u < D; u++) {
                const int k = k0 + u;
                request_lds(k + 1, ktn, k6n);
                real R[7];
                R[0] = buf[u][0].x + kt[0].x; R[1] = buf[u][0].y + kt[0].y; R[2] = buf[u][1].x + kt[1].x; R[3] = buf[u][1].y + kt[1].y;
                R[4] = buf[u][2].x + kt[2].x; R[5] = fma(w5, kt[2].y, buf[u][2].y); R[6] = buf[u][3].x + k6;
                asm volatile("" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]));      // the row is formed BEFORE its buffer is re-requested
                request(k + D, buf[u]);
                real acc = R[6], xn;
                asm volatile("s_nop 4\n\t"
                             LAT_DPP(1, 2, 3, 0) LAT_DPP(1, 2, 4, 1) LAT_DPP(1, 2, 5, 2) LAT_DPP(1, 2, 6, 3) LAT_DPP(1, 2, 7, 4)
                             "s_nop 1\n\t" LAT_MOV " %0, %1\n\t" LAT_DPP(0, 1, 8, 5)
                             : "=&v"(xn), "+v"(acc) : "v"(xr), "v"(R[0]), "v"(R[1]), "v"(R[2]), "v"(R[3]), "v"(R[4]), "v"(R[5]));
                xr = xn;                              // lanes 0..4: x_{k+1}; lane 5: v_k (its Bbar entry is 0)
                *((c < 6 && k < N) ? sRec + LAT_REC * (k < N ? k : 0) + c : sDum + lane) = xn;
#pragma unroll
                for (int q = 0; q < 3; q++) kt[q] = ktn[q];
                k6 = k6n;
            }
        }
    };

    // ---------------- stage-parallel part ----------------
    // Home of the per-row state (slack t, multiplier lambda, second-order term / d-lambda) of this lane's rows:
    //   MEM = false: registers, for the whole solve (NSLOT <= 2: 10 or 13 rows x 3 x NSLOT doubles per lane fit next to the serial passes);
    //   MEM = true : a per-wavefront workspace in global memory (C.lat_ws), one slot at a time through registers, the slot loop ROLLED (no prefetch of the next slot: see `piped`).  At N = 50 (four slots, the
    //                last one two stages deep) the register file cannot hold 156 doubles per lane next to the passes: the compiler's own spilling cost 40 % of an iteration
    //                (serialised scratch reloads at one wavefront per SIMD; 24 k instructions of unrolled slot code against a 64 KB instruction cache).  The explicit home
    //                is read with 16-byte loads issued back to back at the top of a slot visit, (t, lambda) pairs interleaved, the lanes of a wavefront contiguous.
    constexpr int NP = (NR + 1) / 2;
    constexpr size_t WS_TL = (size_t)NR * 64 * 2 * sizeof(real), WS_CR = (size_t)NP * 64 * 2 * sizeof(real), WS_SN = (size_t)64 * 4 * sizeof(real), WS_META = (size_t)64 * 16;
    constexpr size_t WS_SLOT = WS_TL + WS_CR + WS_SN + WS_META;
    static_assert(WS_SLOT <= LAT_WS_SLOT_BYTES, "workspace slot");
    const int nslot = MEM ? (N + 15) >> 4 : NSLOT;
    constexpr int NREG = MEM ? 1 : NSLOT;
    real T[NREG][NR], L[NREG][NR], CR[NREG][NR];
    unsigned amask[NREG], mask_ipm[NREG], nmask[NREG]; real SN[NREG][3];
    // HOME of an instance's row state: the workspace block of the wavefront that STARTED it and its lane group there.  A launch that starts instances is its own home; a
    // resumed instance (hand_mode 2) names its home in the hand-over record
    const int* const hrec_i = O.hand_i ? O.hand_i + (size_t)(valid ? b : 0) * LAT_HAND_I : nullptr;
    const real* const hrec_r = O.hand_r ? O.hand_r + (size_t)(valid ? b : 0) * LAT_HAND_R : nullptr;
    const int hb = resume ? hrec_i[12] : (int)blockIdx.x, hg = resume ? hrec_i[13] : g;
    const int hl = 16 * hg + c;                              // (LPI = 16: this lane's position in the home wavefront)
    char* const wsw = MEM ? C.lat_ws + (size_t)hb * 4 * LAT_WS_SLOT_BYTES : nullptr;
    auto sidx = [&](int j) __attribute__((always_inline)) { return cs + LPI * j; };      // the stage this lane owns in slot j
    auto is_act = [&](int j) __attribute__((always_inline)) { return cs + LPI * j < N; };
    auto for_slots = [&](auto&& body) __attribute__((always_inline)) {
        if constexpr (MEM) {
#pragma unroll 1
            for (int j = 0; j < nslot; j++) body(j);
        } else {
#pragma unroll
            for (int j = 0; j < NSLOT; j++) { body(j); __builtin_amdgcn_sched_barrier(0); }
        }
    };
    auto get_tl = [&](int j, real* Tl, real* Ll) __attribute__((always_inline)) {
        if constexpr (MEM) {
            // (lanes whose stage lies beyond the horizon -- 14 of the 16 lanes of the last slot at N = 50 -- carry t = lambda = 1 and touch no memory: a fifth of the
            // workspace traffic of the kernel was theirs)
            const real2* p = reinterpret_cast<const real2*>(wsw + (size_t)j * LAT_WS_SLOT_BYTES) + hl;
            if (is_act(j)) {
#pragma unroll
                for (int r = 0; r < NR; r++) { const real2 v = p[64 * r]; Tl[r] = v.x; Ll[r] = v.y; }
            } else {
#pragma unroll
                for (int r = 0; r < NR; r++) { Tl[r] = real(1.0); Ll[r] = real(1.0); }
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; r++) { Tl[r] = T[j][r]; Ll[r] = L[j][r]; }
        }
    };
    auto put_tl = [&](int j, const real* Tl, const real* Ll) __attribute__((always_inline)) {
        if constexpr (MEM) {
            real2* p = reinterpret_cast<real2*>(wsw + (size_t)j * LAT_WS_SLOT_BYTES) + hl;
            if (is_act(j)) {
#pragma unroll
                for (int r = 0; r < NR; r++) { real2 v; v.x = Tl[r]; v.y = Ll[r]; p[64 * r] = v; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; r++) { T[j][r] = Tl[r]; L[j][r] = Ll[r]; }
        }
    };
    auto get_cr = [&](int j, real* Cl) __attribute__((always_inline)) {
        if constexpr (MEM) {
            const real2* p = reinterpret_cast<const real2*>(wsw + (size_t)j * LAT_WS_SLOT_BYTES + WS_TL) + hl;
            if (is_act(j)) {
#pragma unroll
                for (int q = 0; q < NP; q++) { const real2 v = p[64 * q]; Cl[2 * q] = v.x; if (2 * q + 1 < NR) Cl[2 * q + 1] = v.y; }
            } else {
#pragma unroll
                for (int r = 0; r < NR; r++) Cl[r] = real(0.0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; r++) Cl[r] = CR[j][r];
        }
    };
    auto put_cr = [&](int j, const real* Cl) __attribute__((always_inline)) {
        if constexpr (MEM) {
            real2* p = reinterpret_cast<real2*>(wsw + (size_t)j * LAT_WS_SLOT_BYTES + WS_TL) + hl;
            if (is_act(j)) {
#pragma unroll
                for (int q = 0; q < NP; q++) { real2 v; v.x = Cl[2 * q]; v.y = 2 * q + 1 < NR ? Cl[2 * q + 1] : real(0.0); p[64 * q] = v; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; r++) CR[j][r] = Cl[r];
        }
    };
    auto get_sn = [&](int j, real* s3) __attribute__((always_inline)) {
        if constexpr (MEM) {
            const real2* p = reinterpret_cast<const real2*>(wsw + (size_t)j * LAT_WS_SLOT_BYTES + WS_TL + WS_CR) + 2 * hl;
            if (is_act(j)) { const real2 a = p[0], b_ = p[1]; s3[0] = a.x; s3[1] = a.y; s3[2] = b_.x; }
            else { s3[0] = real(0.0); s3[1] = real(0.0); s3[2] = real(0.0); }
        } else { s3[0] = SN[j][0]; s3[1] = SN[j][1]; s3[2] = SN[j][2]; }
    };
    auto put_sn = [&](int j, const real* s3) __attribute__((always_inline)) {
        if constexpr (MEM) {
            real2* p = reinterpret_cast<real2*>(wsw + (size_t)j * LAT_WS_SLOT_BYTES + WS_TL + WS_CR) + 2 * hl;
            if (is_act(j)) { real2 a, b_; a.x = s3[0]; a.y = s3[1]; b_.x = s3[2]; b_.y = real(0.0); p[0] = a; p[1] = b_; }
        } else { SN[j][0] = s3[0]; SN[j][1] = s3[1]; SN[j][2] = s3[2]; }
    };
    // working set of the polish, the interior point's set at the hand-over, the set the last polish solve proposes
    struct Meta { unsigned am, mi, nm; };
    auto get_meta = [&](int j) __attribute__((always_inline)) -> Meta {
        Meta m;
        if constexpr (MEM) { m.am = 0u; m.mi = 0u; m.nm = 0u; if (is_act(j)) { const uint4 v = *(reinterpret_cast<const uint4*>(wsw + (size_t)j * LAT_WS_SLOT_BYTES + WS_TL + WS_CR + WS_SN) + hl); m.am = v.x; m.mi = v.y; m.nm = v.z; } }
        else { m.am = amask[j]; m.mi = mask_ipm[j]; m.nm = nmask[j]; }
        return m;
    };
    auto get_meta_if = [&](int j, bool wanted) __attribute__((always_inline)) -> Meta {      // (the interior point never looks at the sets: no load while it runs)
        Meta m; m.am = 0u; m.mi = 0u; m.nm = 0u;
        if (wanted) m = get_meta(j);
        return m;
    };
    auto put_meta = [&](int j, const Meta& m) __attribute__((always_inline)) {
        if constexpr (MEM) { if (is_act(j)) { uint4 v; v.x = m.am; v.y = m.mi; v.z = m.nm; v.w = 0u; *(reinterpret_cast<uint4*>(wsw + (size_t)j * LAT_WS_SLOT_BYTES + WS_TL + WS_CR + WS_SN) + hl) = v; } }
        else { amask[j] = m.am; mask_ipm[j] = m.mi; nmask[j] = m.nm; }
    };
    auto sx_of = [&](int j) __attribute__((always_inline)) { const int s = is_act(j) ? sidx(j) : N - 1; return O.sol_x + (size_t)b * NN * 8 + 8 * (s + 1); };
    auto sg_of = [&](int j) __attribute__((always_inline)) { const int s = is_act(j) ? sidx(j) : N - 1; return O.sol_sigma + ((size_t)b * N + s) * 3; };

    struct StageC { real b[NR], h0[4], h1[4], dts; };
    constexpr bool USE_SPC = MEM || HAND == 2;          // (the lane-contiguous copy of the stage constants: wherever an instance has a home in the workspace)
    const int nslot_h = (N + 15) >> 4;
    auto spc_of = [&](int s) __attribute__((always_inline)) {      // stage s of this instance in its home wavefront's block: slot s / 16, lane 16 hg + s % 16
        return reinterpret_cast<real2*>(C.lat_spc) + ((size_t)hb * nslot_h + (size_t)(s >> 4)) * LAT_SPC_Q * 64 + 16 * hg + (s & 15);
    };
    auto load_consts = [&](int j, StageC& S) __attribute__((always_inline)) {
        int s = is_act(j) ? sidx(j) : N - 1;
        asm volatile("" : "+v"(s));        // opaque per call: these loads are invariant across the interior-point loop, and hoisted out of it they would sit in ~45 registers per slot
        real2 h4[4], g01, g23, dd, rr;
        if constexpr (USE_SPC) {
            const real2* sp = spc_of(s);
#pragma unroll
            for (int i = 0; i < 4; i++) h4[i] = sp[64 * i];
            g01 = sp[64 * 4]; g23 = sp[64 * 5]; dd = sp[64 * 6]; rr = sp[64 * 7];
            S.dts = sp[64 * 8].x;
        } else {
            const real2* cp = reinterpret_cast<const real2*>(Lb + (size_t)s * LATP + 32);
#pragma unroll
            for (int i = 0; i < 4; i++) h4[i] = cp[i];
            g01 = cp[4]; g23 = cp[5]; dd = cp[6]; rr = cp[7];
            S.dts = Lb[(size_t)s * LATP + 48];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { S.h0[i] = h4[i].x; S.h1[i] = h4[i].y; }
        S.b[2] = g01.x; S.b[3] = g01.y; S.b[4] = g23.x; S.b[5] = g23.y;
        S.b[0] = dd.x; S.b[1] = -dd.y; S.b[6] = real(0.0); S.b[7] = real(0.0); S.b[8] = rr.x; S.b[9] = -rr.y;
        if constexpr (WALLS) { const real* w = C.wall_edges + ((size_t)b * N + s) * 2; S.b[10] = w[0]; S.b[11] = -w[1]; S.b[12] = real(0.0); }
    };
    int pmode = 0, pstat = 0, pchecks = 0; bool want_polish = false, skip_second = false, resume_ipm = false;
    // what a slot visit reads before it computes: stage constants, (t, lambda), second-order term / d-lambda, working-set words, eliminated slacks
    struct In { StageC S; real Tl[NR], Ll[NR], Cl[NR]; Meta mt; real s3[3]; };
    constexpr int F_TL = 1, F_CR = 2, F_META = 4, F_SN = 8;
    auto fetch = [&](int j, In& in, int what) __attribute__((always_inline)) {
        load_consts(j, in.S);
        if (what & F_TL) get_tl(j, in.Tl, in.Ll);
        if (what & F_CR) get_cr(j, in.Cl);
        if (what & F_SN) get_sn(j, in.s3);
        in.mt = get_meta_if(j, (what & F_META) && pmode != 0);
    };
    // one pass over the slots (MEM: rolled; the loads of a visit are issued back to back at its top.  Requesting slot j + 1 while slot j computes was measured:
    // barrier terms 48 k -> 40 k cycles per iteration, Newton point unchanged, and the 62-double copy of the prefetched operands per visit ate the gain:
    // 2.46 / 3.42 ms with it, 2.36 / 3.45 ms without)
    // (Round 5, measured and removed: pulling the NEXT slot towards the L2 while this one computes -- three loads of one dword per lane, 128 B apart, touch the 184 cache
    // lines of a slot, nothing copied: 3.05 -> 3.47 ms with walls, 2.19 -> 2.40 without.  A load instruction that touches 64 cache lines costs the memory pipeline more than
    // the round trip it was meant to hide.)
    auto piped = [&](int what, auto&& compute) __attribute__((always_inline)) {
        if constexpr (MEM) {
#pragma unroll 1
            for (int j = 0; j < nslot; j++) { In cur; fetch(j, cur, what); compute(j, cur); }
        } else {
#pragma unroll
            for (int j = 0; j < NSLOT; j++) { In cur; fetch(j, cur, what); compute(j, cur); __builtin_amdgcn_sched_barrier(0); }
        }
    };
    auto slacks = [&](const StageC& S, const real* x, real v, real s1, real s2, real sw, real* out) __attribute__((always_inline)) {
        out[0] = S.b[0] - x[4]; out[1] = x[4] + S.b[1];
#pragma unroll
        for (int i = 0; i < 4; i++) out[2 + i] = S.b[2 + i] - (S.h0[i] * x[0] + S.h1[i] * x[1]) + (i < 2 ? s1 : s2);
        out[6] = s1; out[7] = s2; out[8] = S.b[8] - v; out[9] = v + S.b[9];
        if constexpr (WALLS) { out[10] = S.b[10] - x[3] + sw; out[11] = S.b[11] + x[3] + sw; out[12] = sw; }
    };
    // elimination of the stage-local slacks (each is a leaf of the KKT graph): sigma_g = -(c_g' x + g_g) d_g
    struct Elim { real c10, c11, g1, d1, c20, c21, g2, d2, ch, gh, dh; };
    auto eliminate = [&](const StageC& S, const real* W, const real* ell, Elim& E) __attribute__((always_inline)) {
        E.d1 = lat_rcp(W[2] + W[3] + W[6]); E.d2 = lat_rcp(W[4] + W[5] + W[7]);
        E.c10 = -(W[2] * S.h0[0] + W[3] * S.h0[1]); E.c11 = -(W[2] * S.h1[0] + W[3] * S.h1[1]);
        E.c20 = -(W[4] * S.h0[2] + W[5] * S.h0[3]); E.c21 = -(W[4] * S.h1[2] + W[5] * S.h1[3]);
        E.g1 = C.cp.W_beta * S.dts - ell[2] - ell[3] - ell[6]; E.g2 = C.cp.W_r * S.dts - ell[4] - ell[5] - ell[7];
        if constexpr (WALLS) { E.dh = lat_rcp(W[10] + W[11] + W[12]); E.ch = -(W[10] - W[11]); E.gh = C.wall_weight * S.dts - ell[10] - ell[11] - ell[12]; }
        else { E.dh = real(0.0); E.ch = real(0.0); E.gh = real(0.0); }
    };
    // Active-set polish (the OSQP-style polish of k_solve, without its cold-start rules: here it only ever follows a converged interior point).  pmode = 0 while the
    // interior point of this instance runs, then the round of the polish; amask = the rows of a slot held as EQUALITIES through an augmented Lagrangian
    // (-y t + rho/2 t^2: the shape of a barrier term with W = rho and constant multiplier part y - rho b), every other row absent; y lives in L.
    // working sets tried per polish: three at the hand-over tolerance (of the 3851 N = 50 + walls instances that verify, 2717 / 876 / 242 / 16 do so in round 1 / 2 / 3 / 4),
    // two in the second attempt behind the resumed interior point (the slowest wavefront sets the kernel's time, and these are its instances)
    const int LAT_POLISH_ROUNDS = C.lat_polish_rounds;
    // Penalty of the held rows: polish_rho x lat_rho_scale (fp64: 1e3, i.e. 1e10 at the default 1e7).  A held rate or steering-bound row pins the input of its stage, and
    // the multiplier iteration lambda <- lambda - rho t contracts by S / (S + rho) per pass, S = Rhat + Bbar' P Bbar the curvature of the cost-to-go in that input -- on the
    // open-loop unstable 8 s horizon P grows like exp(2 lambda T), S reaches 1e9..1e12, and at rho = 1e7 the multipliers of such rows stall: 6-7 % of the N = 50 batch ended
    // unverified for that reason alone (PG_RHO sweep, EXPERIMENTS 10.1: unverified 248 / 111 / 53 / 11 of 4096 at 1e7 / 1e9 / 1e10 / 1e11; accuracy against the oracle
    // unchanged up to 1e10, 3x worse with walls at 1e11).  The embedding in k_solve keeps polish_rho as it is (its refinement solves for a correction; tuned at 1e7).
    const real rho = C.polish_rho * C.lat_rho_scale, ptol = C.polish_tol;
    const real BIGP = sizeof(real) == 8 ? real(1e200) : real(1e22);
    // pinned input of a stage under working set `am` (see the header): a held rate row (rows 8, 9 leave the penalty form); kff = the pinned value
    struct Pin { bool on; real kff; };
    auto pin_of = [&](unsigned am, const StageC& S) __attribute__((always_inline)) -> Pin {
        Pin p; p.on = false; p.kff = real(0.0);
        if (!pmode || C.lat_pin == 0) return p;
        if (am & (1u << 8)) { p.on = true; p.kff = S.b[8]; }
        else if (am & (1u << 9)) { p.on = true; p.kff = -S.b[9]; }
        return p;
    };
    // barrier weights of a slot at the current iterate: it = 1/t, W = lambda/t, ell = (sigma mu - corr)/t + lambda - W b
    auto weights = [&](unsigned am, const real* Tl, const real* Ll, const real* Cl, const StageC& S, real sgmu, bool with_corr, real* it_, real* W, real* ell) __attribute__((always_inline)) {
        if (pmode) {
            const Pin pn = pin_of(am, S);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const bool a = ((am >> r) & 1u) && !(pn.on && (r == 8 || r == 9));
                it_[r] = real(1.0); W[r] = a ? rho : real(0.0); ell[r] = a ? Ll[r] - rho * S.b[r] : real(0.0);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            it_[r] = lat_rcp(Tl[r]); W[r] = Ll[r] * it_[r];
            ell[r] = (with_corr ? (sgmu - Cl[r]) * it_[r] : real(0.0)) + Ll[r] - W[r] * S.b[r];
        }
    };
    // every stage-locally eliminated slack needs a pivot: a group without an active row gets its sigma >= 0 row, whose multiplier is then known (the linear cost of the slack)
    auto polish_pivots = [&](unsigned& am, real* Ll, const StageC& S) __attribute__((always_inline)) -> bool {
        bool touched = false;
        if (!(am & 0x04Cu)) { am |= 1u << 6; Ll[6] = C.cp.W_beta * S.dts; touched = true; }
        if (!(am & 0x0B0u)) { am |= 1u << 7; Ll[7] = C.cp.W_r * S.dts; touched = true; }
        if constexpr (WALLS) { if (!(am & 0x1C00u)) { am |= 1u << 12; Ll[12] = C.wall_weight * S.dts; touched = true; } }
        return touched;
    };
    // after a polish solve (tp = slacks at the new point): multiplier update of the held rows and the add / drop decisions of this slot.  Returns the next working set.
    auto polish_rows = [&](bool actj, unsigned am, real* Ll, const real* tp, real ttol, bool& unsettled, const Pin& pn, real gpin) __attribute__((always_inline)) -> unsigned {
        unsigned add = 0u, drop = 0u;
        // the pinned row: its multiplier is minus / plus the gradient of the Lagrangian in the input (the row "ddmax - v" takes the minus sign)
        const int rpin = pn.on ? ((am & (1u << 8)) ? 8 : 9) : -1;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const bool a = (am >> r) & 1u;
            const bool excl = pn.on && (r == 8 || r == 9);      // rows out of the penalty form
            if (r == rpin) Ll[r] = r == 8 ? -gpin : gpin;
            else if (a && !excl) Ll[r] -= rho * tp[r];
            if (actj && a && Ll[r] < real(0.0)) drop |= 1u << r;
            if (actj && a && !(fabs(tp[r]) <= ttol)) unsettled = true;             // written so that a NaN never verifies
            if (actj && !a && !(tp[r] >= -ptol)) add |= 1u << r;
        }
        // at most one NEW row per slack group and round, the more violated one (two rows that share a free slack pin a combination of the states hard)
        if ((add & 0x00Cu) == 0x00Cu) add &= ~(tp[2] <= tp[3] ? (1u << 3) : (1u << 2));
        if ((add & 0x030u) == 0x030u) add &= ~(tp[4] <= tp[5] ? (1u << 5) : (1u << 4));
        if constexpr (WALLS) { if ((add & 0xC00u) == 0xC00u) add &= ~(tp[10] <= tp[11] ? (1u << 11) : (1u << 10)); }
        return (am & ~drop) | add;
    };
    // gradient of the Lagrangian in the pinned input of stage sj at the roll-out's point: F x_s + Bbar'y + S v  (x_s: the node the stage starts from; S = Rhat0 + Bbar'P Bbar)
    auto pin_gradient = [&](int sj, const StageC& S, real v) __attribute__((always_inline)) -> real {
        const real* ax = aux + (size_t)sj * LAT_AUX;
        real g = ax[6] + (real(2.0) * C.cp.R_ddelta * frcp(S.dts) + ax[5]) * v;
        if (sj > 0) {
            const real* xp = sRec + LAT_REC * (sj - 1);
#pragma unroll
            for (int m = 0; m < 5; m++) g += ax[m] * xp[m];
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) g += ax[m] * Q[o.qcurr + 2 + m];
            g += ax[4] * Q[o.ucurr];
        }
        return g;
    };
    // barrier terms of slot j -> stage cost of the Riccati passes (rec[0..5] the matrices, rec[6..11] the vectors)
    auto assemble = [&](int j, In& in, real sgmu, bool matrices) __attribute__((always_inline)) {
        StageC& S = in.S; real* const Tl = in.Tl; real* const Ll = in.Ll; real* const Cl = in.Cl; Meta& mt = in.mt;
        real it_[NR], W[NR], ell[NR]; Elim E;
        if (matrices && pmode) { if (polish_pivots(mt.am, Ll, S)) { put_meta(j, mt); put_tl(j, Tl, Ll); } }
        weights(mt.am, Tl, Ll, Cl, S, sgmu, !matrices, it_, W, ell);
        eliminate(S, W, ell, E);
        real g0 = real(0.0), g1 = real(0.0);
#pragma unroll
        for (int i = 0; i < 4; i++) { g0 += S.h0[i] * ell[2 + i]; g1 += S.h1[i] * ell[2 + i]; }
        if (is_act(j)) {
            real* rec = sRec + LAT_REC * (sidx(j));
            rec[6] = g0 - E.c10 * E.g1 * E.d1 - E.c20 * E.g2 * E.d2;
            rec[7] = g1 - E.c11 * E.g1 * E.d1 - E.c21 * E.g2 * E.d2;
            rec[8] = real(0.0);
            if constexpr (WALLS) rec[9] = (ell[10] - ell[11]) - E.ch * E.gh * E.dh; else rec[9] = real(0.0);
            const Pin pn = pin_of(mt.am, S);
            rec[10] = ell[0] - ell[1];
            rec[11] = pn.on ? -BIGP * pn.kff : ell[8] - ell[9];
            if (matrices) {
                real yy = real(0.0), yr = real(0.0), rr = real(0.0);
#pragma unroll
                for (int i = 0; i < 4; i++) { yy += W[2 + i] * S.h0[i] * S.h0[i]; yr += W[2 + i] * S.h0[i] * S.h1[i]; rr += W[2 + i] * S.h1[i] * S.h1[i]; }
                rec[0] = yy - E.c10 * E.c10 * E.d1 - E.c20 * E.c20 * E.d2;
                rec[1] = yr - E.c10 * E.c11 * E.d1 - E.c20 * E.c21 * E.d2;
                rec[2] = rr - E.c11 * E.c11 * E.d1 - E.c21 * E.c21 * E.d2;
                rec[3] = real(2.0) * C.cp.Q_e * S.dts;
                if constexpr (WALLS) rec[3] += W[10] + W[11] - E.ch * E.ch * E.dh;
                rec[4] = real(2.0) * C.cp.R_delta * S.dts + W[0] + W[1];
                rec[5] = pn.on ? BIGP : real(2.0) * C.cp.R_ddelta * frcp(S.dts) + W[8] + W[9];
            }
        }
    };
    // The vector part of a stage's barrier terms (what the vector pass reads: rec[6], [7], [9], [10], [11]) is LINEAR in ell, and the corrector's
    // ell = (sigma mu - corr) / t + lambda - W b is affine in sigma mu, which is only known after the reduction over the predictor's Newton point.  The pass that forms
    // that point therefore leaves BOTH parts in the stage record -- the constant one in rec[6..11], the coefficient of sigma mu in rec[0..4] (the roll-out result it has
    // just read) -- and a five-multiply-add touch of the record replaces a whole second assembly pass over the row state (N = 50: 2.37 -> 2.19 ms, 3.45 -> 3.12 ms with walls).
    // (fp64 only: split this way, sigma mu / t and -corr / t are rounded separately although they nearly cancel close to the solution -- harmless at 1e-16, but in fp32
    // three more of 512 N = 50 instances missed their tolerance; the fp32 build keeps the separate assembly pass)
    constexpr bool SPLIT_CORR = sizeof(real) == 8;
    auto vec_terms = [&](const StageC& S, const Elim& E, const real* ell, bool with_const, real* out) __attribute__((always_inline)) {
        real g0 = real(0.0), g1 = real(0.0);
#pragma unroll
        for (int i = 0; i < 4; i++) { g0 += S.h0[i] * ell[2 + i]; g1 += S.h1[i] * ell[2 + i]; }
        const real G1 = (with_const ? C.cp.W_beta * S.dts : real(0.0)) - ell[2] - ell[3] - ell[6], G2 = (with_const ? C.cp.W_r * S.dts : real(0.0)) - ell[4] - ell[5] - ell[7];
        out[0] = g0 - E.c10 * G1 * E.d1 - E.c20 * G2 * E.d2;
        out[1] = g1 - E.c11 * G1 * E.d1 - E.c21 * G2 * E.d2;
        if constexpr (WALLS) out[2] = (ell[10] - ell[11]) - E.ch * ((with_const ? C.wall_weight * S.dts : real(0.0)) - ell[10] - ell[11] - ell[12]) * E.dh; else out[2] = real(0.0);
        out[3] = ell[0] - ell[1];
        out[4] = ell[8] - ell[9];
    };
    auto put_vec = [&](int j, const real* qa, const real* qb, const Pin& pn) __attribute__((always_inline)) {
        if (is_act(j)) {
            real* rec = sRec + LAT_REC * (sidx(j));
            rec[6] = qa[0]; rec[7] = qa[1]; rec[8] = real(0.0); rec[9] = qa[2]; rec[10] = qa[3]; rec[11] = pn.on ? -BIGP * pn.kff : qa[4];
            if (!pmode) {       // (a polish has no sigma mu part, and its verdict may still want the point in rec[0..4]: polish_decide stores it as the answer)
#pragma unroll
                for (int i = 0; i < 5; i++) rec[i] = qb[i];
            }
        }
    };
    // Newton point of slot j from the roll-out (x+ of node s+1, v+ of transition s) -> eliminated slacks and the slack of every row
    auto newton = [&](int j, const StageC& S, const Elim& E, real* xn, real& vn, real* sg3, real* tp) __attribute__((always_inline)) {
        const real* rec = sRec + LAT_REC * (is_act(j) ? sidx(j) : N - 1);
#pragma unroll
        for (int m = 0; m < 5; m++) xn[m] = rec[m];
        vn = rec[5];
        sg3[0] = -(E.c10 * xn[0] + E.c11 * xn[1] + E.g1) * E.d1;
        sg3[1] = -(E.c20 * xn[0] + E.c21 * xn[1] + E.g2) * E.d2;
        sg3[2] = WALLS ? -(E.ch * xn[3] + E.gh) * E.dh : real(0.0);
        slacks(S, xn, vn, sg3[0], sg3[1], sg3[2], tp);
    };

    // state of the solve that a hand-over carries from one launch to the next (per instance, the same value in every lane that serves it)
    constexpr int lat_bit[13] = {3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 0, 1, 2};      // local row -> bit of the 16-bit active mask (the numbering of the embedded stage: pigeon_mpc.h)
    const real ntot = (real)(N * NR), intot = real(1.0) / ntot, tol = C.ipm_tol;
    real rp0 = real(0.0), j0 = real(0.0), ms_next = real(0.0), mu0i = C.ipm_mu0, tol_cur = tol, tol_cold = tol, mu = real(0.0), phi = real(1.0);
    int status = PG_MAX_ITER, it = 0, good = 0, wf = 0, trips = 0, trips0 = 0, work = 0;
#ifdef LAT_TRIP_MIX
    unsigned long long mixI = 0, mixP = 0, mixB = 0, mixA = 0;      // experiment: trips of this wavefront with an instance in its interior point / in a polish / both; unfinished instances summed over trips
#endif      // (trips0: trips an instance had behind it when this launch took it over)
    bool done = false, warm = false, warm_try = false, warm_failed = false, deferred = false, warm_tried = false, counted = false;
    const int cap = C.ipm_max_iter;
    __syncthreads();
    if constexpr (XLDS) {      // the stage matrices of this wavefront's instance, once per launch
        const real2* const src = reinterpret_cast<const real2*>(Lb);
        real2* const dst = reinterpret_cast<real2*>(sMat);
#pragma unroll 1
        for (int idx = lane; idx < 16 * N; idx += 64) dst[(idx >> 4) * (LAT_XS / 2) + (idx & 15)] = src[(size_t)(idx >> 4) * (LATP / 2) + (idx & 15)];
        for (int sq = lane; sq < N; sq += 64) { real2 dz; dz.x = Lb[(size_t)sq * LATP + 48]; dz.y = real(0.0); dst[sq * (LAT_XS / 2) + 16] = dz; }
        wave_sync();
    }
    if constexpr (!resume) {
    if constexpr (USE_SPC) {      // file this wavefront's stage constants lane-contiguous (see LAT_SPC_Q): once per launch, read by every slot visit of every pass
#pragma unroll 1
        for (int j = 0; j < nslot_h; j++) {
            if (cs + LPI * j < N) {
                const int s = sidx(j);
                const real2* cp = reinterpret_cast<const real2*>(Lb + (size_t)s * LATP + 32);
                real2* sp = spc_of(s);
#pragma unroll
                for (int q = 0; q < 8; q++) sp[64 * q] = cp[q];
                real2 dz; dz.x = Lb[(size_t)s * LATP + 48]; dz.y = real(0.0); sp[64 * 8] = dz;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // (a lane reads back what it wrote itself -- and, clamped to the last stage, what its neighbour wrote)
    }
    // ---------------- start: v = 0 roll-out (dynamics- and rate-feasible), soft-row slacks just feasible + 1, t = max(slack, tau), lambda = mu0 / t ----------------
    if (frow) forward_pass(false);
    wave_sync();
    stamp(3);
    // an instance whose previous step ended in a solved QP (see "warm start" below)
    warm = C.polish && C.warm_polish && O.solved[b] != 0 && (O.status[b] == PG_SOLVED || O.status[b] == PG_SOLVED_UNVERIFIED);
    // ... whose interior point, should it be needed, starts from the PREVIOUS solution instead of the v = 0 roll-out: slacks of the previous primal point against the new
    // rows, floored at lat_wtau; multipliers the previous ones, floored at lat_wmu / t (every product t lambda >= lat_wmu: a centred neighbourhood of the old optimum)
    const bool wipm = warm && C.lat_wipm != 0 && O.n_list == nullptr;
    // (ms_next: sum t lambda over this lane's rows at the iterate just stored -- the complementarity gap of the next loop top)
    // the damped iterate (x_{s+1}, sigma) of a stage is kept in the output buffers (read-modify-write once per iteration), not in registers
    for_slots([&](int j) __attribute__((always_inline)) {
        const bool actj = is_act(j);
        const int s = actj ? sidx(j) : N - 1;
        real* const SXj = sx_of(j); real* const SGj = sg_of(j);
        StageC S; load_consts(j, S);
        const real* rec = sRec + LAT_REC * s;
        real xs[5], sl[NR];
#pragma unroll
        for (int m = 0; m < 5; m++) xs[m] = wipm ? SXj[2 + m] : rec[m];
        const real vst = wipm ? xs[4] - (SXj - 8)[6] : real(0.0);
        slacks(S, xs, vst, real(0.0), real(0.0), real(0.0), sl);
        const real sig0 = wipm ? C.lat_wtau : real(1.0), tau = wipm ? C.lat_wtau : real(1e-4);
        const real s1 = fmax(wipm ? SGj[0] : real(0.0), fmax(real(0.0), -fmin(sl[2], sl[3])) + sig0), s2 = fmax(wipm ? SGj[1] : real(0.0), fmax(real(0.0), -fmin(sl[4], sl[5])) + sig0);
        real sw = real(0.0);
        if constexpr (WALLS) sw = fmax(wipm ? SGj[2] : real(0.0), fmax(real(0.0), -fmin(sl[10], sl[11])) + sig0);
        slacks(S, xs, vst, s1, s2, sw, sl);
        if (actj && valid) {
            SXj[0] = real(0.0); SXj[1] = C.ux_dummy; SXj[7] = real(0.0);      // the embedded layout pg_get_solution documents: (0, Ux slot, Uy, r, dpsi, e, delta, 0)
#pragma unroll
            for (int m = 0; m < 5; m++) SXj[2 + m] = xs[m];
            SGj[0] = s1; SGj[1] = s2; SGj[2] = sw;
        }
        // cost of the starting point (tracking terms on node s + 1, the linear penalties of the soft-row slacks): it sets the scale of the first barrier parameter below
        if (actj) {
            real j0s = real(0.5) * S.dts * (real(2.0) * C.cp.Q_dpsi * xs[2] * xs[2] + real(2.0) * C.cp.Q_e * xs[3] * xs[3] + real(2.0) * C.cp.R_delta * xs[4] * xs[4])
                       + S.dts * (C.cp.W_beta * s1 + C.cp.W_r * s2 + (WALLS ? C.wall_weight * sw : real(0.0)));
            j0 += j0s;
        }
        real Tl[NR], Ll[NR], Cl[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const real tj = actj ? fmax(sl[r], tau) : real(1.0);
            Tl[r] = tj; Ll[r] = real(1.0); Cl[r] = real(0.0);      // (slots beyond the horizon: t = lambda = 1, never updated, never summed)
            if (actj) rp0 = fmax(rp0, tj - sl[r]);
        }
        put_tl(j, Tl, Ll); put_cr(j, Cl);
        Meta m0; m0.am = 0u; m0.mi = 0u; m0.nm = 0u; put_meta(j, m0);
    });
    rp0 = grp_max<LPI>(rp0);
    // First barrier parameter: ipm_mu0, raised to lat_mu0_cost (10) x (cost of the starting point per row).  The v = 0 roll-out of an open-loop unstable 8 s horizon can start
    // kilometres off the path; from mu = 100 such an instance spends ~12 iterations with step lengths of a few per cent while mu climbs by itself to ~1e5, and it is these
    // instances (one in twenty) that set the kernel's time.  With the scaled start the slowest of the N = 50 batch needs 16 iterations to the hand-over instead of 27
    // and the mean drops from 9.8 to 9.0 (oracle/lat_ipm_numpy.py carries the same rule).
    mu0i = fmax(C.ipm_mu0, C.lat_mu0_cost * grp_sum<LPI>(j0) * intot);
    for_slots([&](int j) __attribute__((always_inline)) {
        real Tl[NR], Ll[NR]; get_tl(j, Tl, Ll);
        const real* Lp = O.lam + ((size_t)b * N + (is_act(j) ? sidx(j) : N - 1)) * 16;
        real ms = real(0.0);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const real lw = wipm ? fmax(fmax(Lp[lat_bit[r]], real(0.0)), C.lat_wmu * lat_rcp(Tl[r])) : mu0i * lat_rcp(Tl[r]);
            Ll[r] = is_act(j) ? lw : real(1.0); ms += Tl[r] * Ll[r];
        }
        ms_next += is_act(j) ? ms : real(0.0);
        put_tl(j, Tl, Ll);
    });
    if (wipm) mu0i = grp_sum<LPI>(ms_next) * intot;
    // with the polish on, the interior point only has to get close enough for the active set to show (polish_ipm_tol); if the polish does not verify from there, the
    // interior point resumes from the centred point (t, mu / t) and runs down to ipm_tol before the polish gets its second and last chance
    // (an instance that starts further than lat_far_cost per row from its optimum's neighbourhood -- the open-loop roll-out of an unstable horizon, kilometres off the
    // path -- almost never verifies at the hand-over tolerance: of the N = 50 batch 107 of 140 such instances do not, against 17 of 3956 of the rest.  It goes straight
    // down to ipm_tol and has the one polish behind it.)
    tol_cur = (C.polish && C.polish_ipm_tol > tol && !(grp_sum<LPI>(j0) * intot > C.lat_far_cost)) ? C.polish_ipm_tol : tol;
    // Warm start of the ACTIVE SET (the reference runs the lateral QP with OSQP's WarmStart = true, decoupled_lat_long.jl:139, inside the same 100 Hz loop as the coupled
    // one, Pigeon.jl:34 / model_predictive_control.jl:80-100).  An instance whose previous step ended in a solved QP first tries the polish from that step's working set
    // and multipliers on the NEW QP data: a verified round IS the exact optimum of the new QP whatever the guess was.  The cold start above has been prepared anyway (its
    // multipliers wait in the second-order slot, where a hand-over to the polish would put them): a warm attempt that does not verify within lat_warm_rounds working sets
    // resumes as the cold interior point, exactly as a failed first polish resumes the interior point it interrupted.
    // Back-off: the instances a warm attempt does not serve are mostly the same ones from step to step (far horizons whose working set turns over by a dozen rows per
    // 10 ms), and a failed attempt costs its rounds ON TOP of the cold solve -- in a kernel that ends with its slowest instance.  An instance whose attempt failed skips the
    // next 1, 3, 7, 15, 31 attempts (level in bits 8.., remaining skips in bits 0..7 of wfail[b]); a verified attempt clears the word.
    wf = O.wfail ? O.wfail[b] : 0;
    warm_try = warm && C.lat_warm_rounds > 0 && (wf & 0xFF) == 0 && !listm;
    warm_tried = warm_try;
    tol_cold = tol_cur;
    if (warm_try) {
        for_slots([&](int j) __attribute__((always_inline)) {
            real Tl[NR], Ll[NR]; get_tl(j, Tl, Ll);
            put_cr(j, Ll);
            const int s = is_act(j) ? sidx(j) : N - 1;
            const unsigned pm = is_act(j) ? (unsigned)O.active[(size_t)b * N + s] : 0u;
            const real* Lp = O.lam + ((size_t)b * N + s) * 16;
            unsigned mk = 0u;
#pragma unroll
            for (int r = 0; r < NR; r++) { const bool a = (pm >> lat_bit[r]) & 1u; mk |= a ? (1u << r) : 0u; Ll[r] = a ? Lp[lat_bit[r]] : real(0.0); }
            put_tl(j, Tl, Ll);
            Meta m; m.am = mk; m.mi = mk; m.nm = mk; put_meta(j, m);
        });
        pmode = 1; pchecks = 0; status = PG_SOLVED;
    }
    if (defer && !warm_try) { deferred = true; done = true; }      // (first of two launches: this instance has no warm attempt to make -- cold, or backing off: the second launch solves it)
    } else {
        // ---------------- resume (hand_mode 2): the scalars of the solve from the hand-over record, its row state from (or in) its home ----------------
        {
            mu = hrec_r[0]; phi = hrec_r[1]; rp0 = hrec_r[2]; mu0i = hrec_r[3]; tol_cur = hrec_r[4]; tol_cold = hrec_r[5];
            it = hrec_i[0]; good = hrec_i[1]; status = hrec_i[2]; pmode = hrec_i[3]; pstat = hrec_i[4]; pchecks = hrec_i[5];
            const int fl = hrec_i[6];
            want_polish = (fl & 1) != 0; resume_ipm = (fl & 2) != 0; warm_try = (fl & 4) != 0; warm_failed = (fl & 8) != 0; warm_tried = (fl & 16) != 0; warm = (fl & 32) != 0;
            wf = hrec_i[7]; trips = hrec_i[8]; trips0 = trips;
        }
        if constexpr (LPI == 64) {
            // one instance per wavefront: lane cs takes stage cs of the home layout (slot cs / 16, lane 16 hg + cs % 16 of the home wavefront) into its registers, for good
            const char* hw = C.lat_ws + ((size_t)hb * 4 + (size_t)(cs >> 4)) * LAT_WS_SLOT_BYTES;
            const int hl64 = 16 * hg + (cs & 15);
            if (is_act(0)) {
                const real2* pt = reinterpret_cast<const real2*>(hw) + hl64;
#pragma unroll
                for (int r = 0; r < NR; r++) { const real2 v = pt[64 * r]; T[0][r] = v.x; L[0][r] = v.y; }
                const real2* pc = reinterpret_cast<const real2*>(hw + WS_TL) + hl64;
#pragma unroll
                for (int q = 0; q < NP; q++) { const real2 v = pc[64 * q]; CR[0][2 * q] = v.x; if (2 * q + 1 < NR) CR[0][2 * q + 1] = v.y; }
                const real2* ps = reinterpret_cast<const real2*>(hw + WS_TL + WS_CR) + 2 * hl64;
                const real2 sa = ps[0], sb = ps[1]; SN[0][0] = sa.x; SN[0][1] = sa.y; SN[0][2] = sb.x;
                const uint4 mv = *(reinterpret_cast<const uint4*>(hw + WS_TL + WS_CR + WS_SN) + hl64);
                amask[0] = mv.x; mask_ipm[0] = mv.y; nmask[0] = mv.z;
            } else {
#pragma unroll
                for (int r = 0; r < NR; r++) { T[0][r] = real(1.0); L[0][r] = real(1.0); CR[0][r] = real(0.0); }
                SN[0][0] = real(0.0); SN[0][1] = real(0.0); SN[0][2] = real(0.0); amask[0] = 0u; mask_ipm[0] = 0u; nmask[0] = 0u;
            }
        }
        for_slots([&](int j) __attribute__((always_inline)) {      // the complementarity gap of the iterate at hand (what the update pass of the last trip left in ms_next)
            real Tl[NR], Ll[NR]; get_tl(j, Tl, Ll);
            real ms = real(0.0);
#pragma unroll
            for (int r = 0; r < NR; r++) ms += Tl[r] * Ll[r];
            ms_next += is_act(j) ? ms : real(0.0);
        });
    }

    // the verdict of a polish solve for this instance, from the per-slot results of polish_rows (the proposed sets are in the slots' meta words, the eliminated slacks
    // of the solve in their sn words): verified (the point is primal and dual feasible: a KKT point of the full QP, stored as the answer), refine (same set, held rows
    // not yet at t = 0: the next solve starts from the updated multipliers), or a new working set.
    // Returns true when the working set changed (the corrector half of this iteration then carries nothing for this instance: its vector pass has the old gains).
    unsigned long long dbg_tr0 = 0ull, dbg_tr1 = 0ull; int dbg_n = 0;      // diagnostic launch (prof != nullptr): one record per polish verdict of this instance
    auto polish_decide = [&](bool unsettled_, bool second_half) __attribute__((always_inline)) -> bool {
        real chg = real(0.0), nch = real(0.0);
        for_slots([&](int j) __attribute__((always_inline)) { const Meta m = get_meta(j); chg = fmax(chg, (is_act(j) && m.nm != m.am) ? real(1.0) : real(0.0)); nch += is_act(j) ? (real)__popc(m.nm ^ m.am) : real(0.0); });
        const real nchg = grp_sum<LPI>(nch);          // rows that enter or leave the working set in this verdict
        const bool conv = !(grp_max<LPI>(unsettled_ ? real(1.0) : real(0.0)) > real(0.0));
        // decisions wait for settled multipliers (lat_settle: 1 = warm attempts, 2 = every polish): on an open-loop unstable horizon a held row that is still 1e-5 off its
        // bound moves the far end of the trajectory by metres, and the rows that then LOOK violated send the working set off (traces: 1 -> 15 -> 90 rows changing per round)
        const bool changed = grp_max<LPI>(chg) > real(0.0) && (conv || !(C.lat_settle == 2 || (C.lat_settle == 1 && warm_try)));
        if (prof) {      // nibble: changed | settled << 1 | warm attempt << 2 | second half << 3; byte: rows that change
            const int nc = (int)fmin(nchg, real(255.0));
            if (dbg_n < 16) dbg_tr0 |= (unsigned long long)((changed ? 1 : 0) | (conv ? 2 : 0) | (warm_try ? 4 : 0) | (second_half ? 8 : 0)) << (4 * dbg_n);
            if (dbg_n < 8) dbg_tr1 |= (unsigned long long)nc << (8 * dbg_n);
            dbg_n++;
        }
        pchecks++;
        if (!changed && conv) {
            for_slots([&](int j) __attribute__((always_inline)) {
                if (is_act(j) && valid) {
                    const real* rec = sRec + LAT_REC * (sidx(j));
                    real* const SXj = sx_of(j); real* const SGj = sg_of(j);
                    real s3[3]; get_sn(j, s3);
#pragma unroll
                    for (int m = 0; m < 5; m++) SXj[2 + m] = rec[m];
                    SGj[0] = s3[0]; SGj[1] = s3[1]; SGj[2] = s3[2];
                }
            });
            pstat = pmode; done = true;
            return false;
        }
        if (changed) {
            for_slots([&](int j) __attribute__((always_inline)) {
                Meta m = get_meta(j); m.am = m.nm; put_meta(j, m);
                real Tl[NR], Ll[NR]; get_tl(j, Tl, Ll);
#pragma unroll
                for (int r = 0; r < NR; r++) Ll[r] = ((m.am >> r) & 1u) ? Ll[r] : real(0.0);
                put_tl(j, Tl, Ll);
            });
            pmode++;
        }
        // (a warm attempt: lat_warm_rounds working sets, two more while the set moves by a row or two per round -- a settled loop whose set shifts by one stage gets there,
        //  and ONE instance that does not costs the launch a cold solve; an attempt that turns over a dozen rows is on its way out and is left to the cold list at once)
        const int round_cap = warm_try ? C.lat_warm_rounds + (nchg <= real(2.0) ? 2 : 0) : (tol_cur > tol ? LAT_POLISH_ROUNDS : LAT_POLISH_ROUNDS - 1);
        if (pmode > round_cap || pchecks > 2 * round_cap + 1) {
            pstat = -1;
            // resume the interior point (takes effect at the top of the next iteration: the rest of this one still belongs to the polish), or -- second failure --
            // nothing verified: the interior-point iterate stands.  (Both flags are assigned on both paths: "if (again) resume = true; else done = true" is folded by the
            // optimiser into ONE store through a selected pointer into the lambda's capture block, which then -- with every captured variable -- lives in scratch memory:
            // the kernel ran twice as long.)
            const bool wdefer = warm_try && defer;             // (first of two launches: a warm attempt that did not verify leaves the instance to the second one)
            const bool again = !wdefer && (warm_try || tol_cur > tol);      // (one launch: a warm attempt that did not verify goes on to the cold start)
            resume_ipm = again; done = done || !again; deferred = deferred || wdefer; warm_failed = warm_failed || wdefer;
        }
        return changed;
    };

    bool need_a1 = true;
    while (true) {
        {   // complementarity gap and the stopping rules (per instance = per row of lanes)
            const real mu_new = grp_sum<LPI>(ms_next) * intot;
            if (resume_ipm) {       // the polish at the hand-over tolerance did not verify: the interior point resumes where it stopped (t is untouched by the polish,
                                    // lambda was set aside) and goes all the way down before the polish gets its second and last chance
                // (or the warm attempt did not verify: the cold start prepared before it takes over, at the tolerance a cold instance starts with)
                resume_ipm = false; tol_cur = warm_try ? tol_cold : tol; pstat = warm_try ? 0 : pstat; warm_failed = warm_failed || warm_try; warm_try = false; pmode = 0; status = PG_MAX_ITER; need_a1 = true;
                real msr = real(0.0);
                for_slots([&](int j) __attribute__((always_inline)) {
                    real Tl[NR], Ll[NR], Cl[NR]; get_tl(j, Tl, Ll); get_cr(j, Cl);
                    real ms = real(0.0);
#pragma unroll
                    for (int r = 0; r < NR; r++) { Ll[r] = is_act(j) ? Cl[r] : real(1.0); ms += Tl[r] * Ll[r]; }
                    msr += is_act(j) ? ms : real(0.0);
                    put_tl(j, Tl, Ll);
                });
                mu = grp_sum<LPI>(msr) * intot;
            } else if (!done && !pmode && !want_polish) {
                mu = mu_new;
                // (convergence is tested BEFORE the cap: an instance that meets its tolerances exactly at the cap is solved, not PG_MAX_ITER)
                if (!(mu == mu) || fabs(mu) > PG_BIG) { status = PG_NUMERICAL; done = true; }
                else if (mu <= tol_cur && phi * fmax(rp0, real(1.0)) <= tol_cur) { status = PG_SOLVED; if (C.polish && (tol_cur > tol || C.lat_polish2)) want_polish = true; else done = true; }
                else if (it >= cap && !(cap >= 20 && good >= 3 && it < cap + 20)) done = true;               // iteration cap (a converging attempt gets twenty more)
            }
            if (want_polish) {      // the interior point has converged: the rows with lambda > t are handed to the polish as its first working set, with their multipliers
                want_polish = false; pmode = 1; pchecks = 0; status = PG_SOLVED; need_a1 = true;
                for_slots([&](int j) __attribute__((always_inline)) {
                    real Tl[NR], Ll[NR]; get_tl(j, Tl, Ll);
                    unsigned mk = 0u;
#pragma unroll
                    for (int r = 0; r < NR; r++) if (is_act(j) && Ll[r] > Tl[r]) mk |= 1u << r;
                    put_cr(j, Ll);          // the interior point's multipliers wait in the (idle) second-order slot: a polish that does not verify hands them back
#pragma unroll
                    for (int r = 0; r < NR; r++) Ll[r] = ((mk >> r) & 1u) ? Ll[r] : real(0.0);
                    put_tl(j, Tl, Ll);
                    Meta m; m.am = mk; m.mi = mk; m.nm = mk; put_meta(j, m);
                });
            }
        }
        if (__all(done)) break;
        stamp(5);
        // ---- predictor: sigma = 0, no second-order term ----
        // (its barrier terms were assembled by the update pass of the previous iteration, which had t and lambda in its registers anyway; only the first iteration and
        // an instance whose mode has just changed -- hand-over to the polish, resumed interior point -- assemble here)
        if (need_a1) piped(F_TL | F_META, [&](int j, In& in) __attribute__((always_inline)) { assemble(j, in, real(0.0), true); });
        need_a1 = false;
        wave_sync();
        stamp(0);
        // (round 6: lat_aux is read by the polish check of a PINNED stage only -- a wavefront none of whose instances is in its polish runs the copy of the two passes
        //  without those stores: one copy or the other per pass, no test per store (that was measured slower in round 5))
        const bool aux_need = C.lat_aux_gate == 0 || __any(pmode != 0);
#ifdef LAT_TRIP_MIX
        { const bool ai = __any(!done && pmode == 0 && valid), ap_ = __any(!done && pmode != 0 && valid); mixI += ai ? 1 : 0; mixP += ap_ ? 1 : 0; mixB += (ai && ap_) ? 1 : 0; mixA += __popcll(__ballot(!done && valid && cs == 0)); }
#endif
        if (frow) { if (aux_need) matrix_pass(std::true_type{}); else matrix_pass(std::false_type{}); }
        wave_sync();
        stamp(1);
        if (frow) forward_pass(true);
        wave_sync();
        stamp(3);
        real rmax = real(0.0), S2 = real(0.0); bool unsettled = false;
        piped(F_TL | F_META, [&](int j, In& in) __attribute__((always_inline)) {
            StageC& S = in.S; real* const Tl = in.Tl; real* const Ll = in.Ll; real* const Cl = in.Cl; Meta& mt = in.mt;
            real it_[NR], W[NR], ell[NR], tp[NR], xn[5], vn, sg3[3]; Elim E;
            weights(mt.am, Tl, Ll, Cl, S, real(0.0), false, it_, W, ell);
            eliminate(S, W, ell, E);
            newton(j, S, E, xn, vn, sg3, tp);
            const Pin pn = pin_of(mt.am, S);
            real qa[5], qb[5];
            if (pmode) {
                // (!done: an instance whose polish has verified keeps its lanes in the loop while the other instances of the wavefront go on -- its multipliers must not take
                //  further steps of the multiplier iteration meanwhile: how many trips that is depends on WHO shares the wavefront, and in a list-mode launch that is the order
                //  in which the to-do list was filled.  Round 6: found by the bit-for-bit test of the lateral closed loop -- the solutions were the same, the multipliers
                //  handed to the next step's warm attempt 1e-14 apart, the loop 1e-7 apart after ten steps)
                if (!done) {
                    const real gpin = (pn.on && is_act(j)) ? pin_gradient(sidx(j), S, vn) : real(0.0);
                    put_sn(j, sg3); mt.nm = polish_rows(is_act(j), mt.am, Ll, tp, real(0.01) * ptol, unsettled, pn, gpin); put_meta(j, mt); put_tl(j, Tl, Ll);
                }
                if constexpr (SPLIT_CORR) {      // the refinement solve behind this one: same set, the multipliers just updated (no sigma mu in a polish)
                    weights(mt.am, Tl, Ll, Cl, S, real(0.0), false, it_, W, ell);
                    vec_terms(S, E, ell, true, qa);
#pragma unroll
                    for (int i = 0; i < 5; i++) qb[i] = real(0.0);
                }
            }
            else {
                real rm = real(0.0), s2 = real(0.0);      // (one select per visit, not per row: a wavefront at one wave per SIMD pays for every instruction it issues)
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const real dt_ = tp[r] - Tl[r], dl_ = -W[r] * tp[r];
                    Cl[r] = dt_ * dl_;
                    // step to the boundary: -dt/t = 1 - tp/t and, for the affine direction, -dl/lambda = tp/t
                    const real rho_ = tp[r] * it_[r];
                    rm = fmax(rm, fmax(real(1.0) - rho_, rho_));
                    s2 += Cl[r];
                }
                rmax = fmax(rmax, is_act(j) ? rm : real(0.0)); S2 += is_act(j) ? s2 : real(0.0);
                put_cr(j, Cl);
                if constexpr (SPLIT_CORR) {      // the corrector's vector terms: ell = (ell at sigma mu = 0) + sigma mu / t
#pragma unroll
                    for (int r = 0; r < NR; r++) ell[r] = ell[r] - Cl[r] * it_[r];
                    vec_terms(S, E, ell, true, qa);
                    vec_terms(S, E, it_, false, qb);
                }
            }
            if constexpr (SPLIT_CORR) put_vec(j, qa, qb, pn);
        });
        skip_second = false;
        if (pmode && !done) skip_second = polish_decide(unsettled, false);
        rmax = grp_max<LPI>(rmax); S2 = grp_sum<LPI>(S2);
        const real aaff = rmax > real(1.0) ? frcp(rmax) : real(1.0);
        // rounding floor: once mu is within 1e4 x of the tolerance and the affine direction can no longer move, further iterations only add noise
        if (!done && !pmode && mu <= real(1e4) * tol && aaff < real(0.3) && phi * fmax(rp0, real(1.0)) <= tol) { status = PG_SOLVED; if (C.polish && (tol_cur > tol || C.lat_polish2)) want_polish = true; else done = true; }
        // sum (t + a dt)(lambda + a dl) = (1 - a) sum t lambda + a^2 sum dt dl   (t dl + lambda dt = -t lambda for the affine direction)
        const real mu_aff = (real(1.0) - aaff) * mu + aaff * aaff * S2 * intot;
        real sg = fmin(mu_aff * frcp(mu), real(1.0)); sg = sg * sg * sg;
        const real sgmu = sg * mu;
        stamp(4);
        // ---- corrector ----
        if constexpr (!SPLIT_CORR) piped(F_TL | F_CR | F_META, [&](int j, In& in) __attribute__((always_inline)) { assemble(j, in, sgmu, false); });
        else for_slots([&](int j) __attribute__((always_inline)) {
            if (is_act(j)) {
                real* rec = sRec + LAT_REC * (sidx(j));
                constexpr int dst[5] = {6, 7, 9, 10, 11};
#pragma unroll
                for (int i = 0; i < 5; i++) { const real v = rec[dst[i]], w = rec[i]; rec[dst[i]] = pmode ? v : fma(sgmu, w, v); }      // (a select: sigma mu of an instance in its polish is not a number)
            }
        });
        wave_sync();
        stamp(0);
        if (frow) { if (aux_need) vector_pass(std::true_type{}); else vector_pass(std::false_type{}); }
        wave_sync();
        stamp(2);
        if (frow) forward_pass(true);
        wave_sync();
        stamp(3);
        real T1 = real(0.0), T2 = real(0.0);
        rmax = real(0.0); unsettled = false;
        piped(F_TL | F_CR | F_META, [&](int j, In& in) __attribute__((always_inline)) {
            StageC& S = in.S; real* const Tl = in.Tl; real* const Ll = in.Ll; real* const Cl = in.Cl; Meta& mt = in.mt;
            real it_[NR], W[NR], ell[NR], tp[NR], xn[5], vn, sg3[3]; Elim E;
            weights(mt.am, Tl, Ll, Cl, S, sgmu, true, it_, W, ell);
            eliminate(S, W, ell, E);
            newton(j, S, E, xn, vn, sg3, tp);
            if (!(pmode && done)) put_sn(j, sg3);
            if (pmode) { if (!skip_second && !resume_ipm && !done) {
                const Pin pn = pin_of(mt.am, S);
                const real gpin = (pn.on && is_act(j)) ? pin_gradient(sidx(j), S, vn) : real(0.0);
                mt.nm = polish_rows(is_act(j), mt.am, Ll, tp, ptol, unsettled, pn, gpin); put_meta(j, mt); put_tl(j, Tl, Ll); } }
            else {
                real rm = real(0.0), t1 = real(0.0), t2 = real(0.0);
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const real dt_ = tp[r] - Tl[r], dl_ = (sgmu - Cl[r]) * it_[r] - W[r] * tp[r];
                    Cl[r] = dl_;                            // (the second-order term has done its job: the slot now carries d-lambda to the update below)
                    rm = fmax(rm, fmax(-dt_ * it_[r], -dl_ * lat_rcp(Ll[r])));
                    t1 += Tl[r] * dl_ + Ll[r] * dt_;
                    t2 += dt_ * dl_;
                }
                rmax = fmax(rmax, is_act(j) ? rm : real(0.0)); T1 += is_act(j) ? t1 : real(0.0); T2 += is_act(j) ? t2 : real(0.0);
                put_cr(j, Cl);
            }
        });
        if (pmode && !done && !skip_second && !resume_ipm) (void)polish_decide(unsettled, true);
        rmax = grp_max<LPI>(rmax); T1 = grp_sum<LPI>(T1); T2 = grp_sum<LPI>(T2);
        const real alpha = rmax > real(0.995) ? real(0.995) * frcp(rmax) : real(1.0);
        // rounding floor, second form: a step that would MULTIPLY mu near the tolerance is a Newton direction computed at a conditioning the arithmetic no longer
        // carries -- the iterate at hand is as good as it gets
        if (!done && !pmode && !want_polish && mu <= (sizeof(real) == 8 ? real(1e5) : real(1e2)) * tol && phi * fmax(rp0, real(1.0)) <= tol) {
            const real mnew = mu + (alpha * T1 + alpha * alpha * T2) * intot;
            if (!(mnew <= real(4.0) * mu)) { status = PG_SOLVED; if (C.polish && (tol_cur > tol || C.lat_polish2)) want_polish = true; else done = true; }
        }
        const bool ipm_on = !done && !pmode && !want_polish;
        const real a = ipm_on ? alpha : real(0.0);
        ms_next = real(0.0);
        piped(F_TL | F_CR | F_SN | F_META, [&](int j, In& in) __attribute__((always_inline)) {
            const bool actj = is_act(j);
            StageC& S = in.S; real* const Tl = in.Tl; real* const Ll = in.Ll; real* const Cl = in.Cl; real* const s3 = in.s3;
            const real* rec = sRec + LAT_REC * (actj ? sidx(j) : N - 1);
            real xn[5], tp[NR];
#pragma unroll
            for (int m = 0; m < 5; m++) xn[m] = rec[m];
            slacks(S, xn, rec[5], s3[0], s3[1], s3[2], tp);
            real ms = real(0.0);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                Tl[r] += (actj && ipm_on) ? a * (tp[r] - Tl[r]) : real(0.0);       // (a select, not a * 0: an instance in its polish has no meaningful tp here)
                Ll[r] += (actj && ipm_on) ? a * Cl[r] : real(0.0);
                ms += Tl[r] * Ll[r];
            }
            ms_next += actj ? ms : real(0.0);
            if (ipm_on) put_tl(j, Tl, Ll);
            if (actj && valid && ipm_on) {
                real* const SXj = sx_of(j); real* const SGj = sg_of(j);
#pragma unroll
                for (int m = 0; m < 5; m++) { const real cur = SXj[2 + m]; SXj[2 + m] = cur + a * (xn[m] - cur); }
#pragma unroll
                for (int m = 0; m < 3; m++) { const real cur = SGj[m]; SGj[m] = cur + a * (s3[m] - cur); }
            }
            assemble(j, in, real(0.0), true);       // the next predictor's barrier terms at the iterate just formed (its roll-out results in rec[0..5] have been read above)
        });
        if (ipm_on) {
            phi *= (real(1.0) - alpha);
            good = alpha > real(0.5) ? good + 1 : 0;
            if (mu > real(1e8) * mu0i) done = true;           // diverging: give up (PG_MAX_ITER)
            it++;
        }
        wave_sync();
        stamp(4);
        trips++;
        if constexpr (HAND == 1) {
            // Hand-over (see the header of this kernel).  Every instance that finishes is counted once on the device; at a trip boundary a wavefront with unfinished
            // instances stops when the batch has at most hand_target of them left (not before hand_min trips), or after hand_cap trips.
            if (done && !counted && valid && cs == 0) atomicAdd(O.hand_done, 1);
            counted = counted || done;
            const int fin = __builtin_amdgcn_readfirstlane(__hip_atomic_load(O.hand_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            // (hand_work: a DETERMINISTIC stand-in for "at a fixed time" -- a trip costs a wavefront a fixed part (the serial passes serve its four instances at once) and a part
            //  per unfinished instance (the stage-parallel visits): the wavefront stops when hand_w0 + #unfinished, summed over its trips, reaches hand_work)
            work += O.hand_w0 + __popcll(__ballot(!done && valid && cs == 0));
            const bool go = (O.hand_cap > 0 && trips >= O.hand_cap) || (O.hand_target > 0 && trips >= O.hand_min && B - fin <= O.hand_target) || (O.hand_work > 0 && work >= O.hand_work);
            if (go && !__all(done)) {
                if (!done) {
                    if (valid && cs == 0) {
                        real* hr = O.hand_r + (size_t)b * LAT_HAND_R; int* hi = O.hand_i + (size_t)b * LAT_HAND_I;
                        hr[0] = mu; hr[1] = phi; hr[2] = rp0; hr[3] = mu0i; hr[4] = tol_cur; hr[5] = tol_cold;
                        hi[0] = it; hi[1] = good; hi[2] = status; hi[3] = pmode; hi[4] = pstat; hi[5] = pchecks;
                        hi[6] = (want_polish ? 1 : 0) | (resume_ipm ? 2 : 0) | (warm_try ? 4 : 0) | (warm_failed ? 8 : 0) | (warm_tried ? 16 : 0) | (warm ? 32 : 0);
                        hi[7] = wf; hi[8] = trips; hi[12] = hb; hi[13] = hg;
                    }
                    deferred = true;
                }
                break;
            }
        }
    }
    stamp(5);
    if (prof && valid && cs == 0) for (int i = 0; i < 6; i++) prof[(size_t)b * 6 + i] = pc[i];
    if (prof && valid && cs == 0) { unsigned long long* tl = prof + (size_t)B * 6 + 1024 + (size_t)b * 3; tl[0] = dbg_tr0; tl[1] = dbg_tr1;
#ifdef LAT_MP_TIMING
        tl[0] = (mp[0] >> 4) | ((mp[1] >> 4) << 32); tl[1] = (mp[2] >> 4) | ((mp[3] >> 4) << 32);
#endif
#ifdef LAT_TRIP_MIX
        tl[0] = mixI | (mixP << 16) | (mixB << 32) | (mixA << 48);
#endif
        tl[2] = (unsigned long long)dbg_n | ((unsigned long long)trips << 32) | ((unsigned long long)trips0 << 48); }      // (the [B][3] region k_solve uses for its timeline)

    // ---------------- outputs ----------------
    if (valid && cs == 0) {
        if constexpr (HAND == 1) { if (deferred) { if (pmode != 0) O.todo[B - 1 - atomicAdd(O.n_todo + 2, 1)] = b; else O.todo[atomicAdd(O.n_todo, 1)] = b; } }
        else if (deferred) O.todo[atomicAdd(O.n_todo, 1)] = b;
        if (O.wfail && !listm) {        // (the back-off word belongs to the launch that makes -- or skips -- the warm attempt)
            if constexpr (!resume) wf = O.wfail[b];      // (read AGAIN rather than kept in a register across the whole solve)
            const int lvl = (wf >> 8) & 0xFF, nl = lvl < 5 ? lvl + 1 : 5;
            O.wfail[b] = !warm ? 0 : (warm_tried ? (warm_failed ? ((nl << 8) | ((1 << nl) - 1)) : 0) : ((lvl << 8) | (((wf & 0xFF) > 0 ? (wf & 0xFF) - 1 : 0))));
        }
    }
    if (valid && !deferred) {
        real* SX = O.sol_x + (size_t)b * NN * 8;
        if (cs < 8) SX[cs] = cs == 1 ? C.ux_dummy : (cs >= 2 && cs < 6 ? Q[o.qcurr + cs] : (cs == 6 ? Q[o.ucurr] : real(0.0)));
        for_slots([&](int j) __attribute__((always_inline)) {
            if (is_act(j)) {
                real Tl[NR], Ll[NR], Cl[NR]; get_tl(j, Tl, Ll); get_cr(j, Cl);
                const Meta m = get_meta(j);
                unsigned mask = 0;
                real* const Lst = O.lam + ((size_t)b * N + sidx(j)) * 16;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const bool on = pstat > 0 ? ((m.am >> r) & 1u) : (pmode ? ((m.mi >> r) & 1u) : (Ll[r] > Tl[r]));      // the polish's verified set / the interior point's at hand-over
                    if (on) mask |= 1u << lat_bit[r];
                    // multipliers for the next step's warm attempt: the verified set's, else the interior point's (set aside in the second-order slot at the hand-over)
                    Lst[lat_bit[r]] = pstat > 0 ? Ll[r] : (pmode ? Cl[r] : Ll[r]);      // (indexed like the mask bits: pg_get_multipliers)
                }
                O.active[(size_t)b * N + sidx(j)] = (uint16_t)mask;
            }
        });
        if (cs == 0) {
            // get_next_control (decoupled_lat_long.jl:275-278): delta of node 2 from the QP, Fx of the seeded node 2
            const real d = (O.sol_x + (size_t)b * NN * 8 + 8)[6] * C.un0, Fx = nodes[((size_t)b * NN + 1) * 10 + 7];
            real* U = O.u_out + (size_t)b * 3;
            U[0] = d; U[1] = Fx > real(0.0) ? Fx * C.veh.fwd_frac : Fx * C.veh.fwb_frac; U[2] = Fx > real(0.0) ? Fx * C.veh.rwd_frac : Fx * C.veh.rwb_frac;
            O.status[b] = (status == PG_SOLVED && C.polish && pstat < 0) ? PG_SOLVED_UNVERIFIED : status; O.iters[b] = it; O.mu[b] = mu; O.polish[b] = pstat;
            O.solved[b] = 1;      // model_predictive_control.jl:76: solved = true
        }
    }
}
