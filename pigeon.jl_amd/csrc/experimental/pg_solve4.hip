// k_solve4: the interior-point method of k_solve (same Newton systems, same Mehrotra logic, same acceptance rules) re-laid-out for lane
// efficiency.  Included by pg_kernels.hip inside namespace pg.
//
// k_solve gives a whole wavefront to one instance; its Riccati vector passes and roll-outs keep 8 of 64 lanes busy and its stage phases 30
// of 64, and the 33 KB of LDS per instance caps a CU at four instances in flight.  Here a wavefront carries FOUR instances, sixteen lanes
// each: lane (g, r, h) = 16 g + 2 r + h owns row r, column half h (columns 4h .. 4h+3) of the 8x8 Riccati matrix of instance g, component r
// of its vectors, and stages gl, gl + 16 (gl = 2 r + h) of its stage-parallel phases.  What lives where:
//   registers   the barrier iterate (t, lambda, corr, 1/t) of the lane's stages, the Riccati matrix row-half, the vector component
//   LDS         a 3-slot ring of stage blocks streamed from L2 (prefetch distance 3 stages), one exchange buffer per instance (the
//               all-gather of P [Abar | Bbar | cbar | p] rows, re-used for the symmetrisation transpose), the per-stage results of the
//               matrix pass (K, S^-1, kff, P c): 30 doubles per stage.  9.6 KB per instance at N = 30 -> four waves (16 instances) per CU.
//   global (L2) the assembled stage costs (Qhat, qhat, Rhat, rhat: 24 doubles per node, prefetched one stage ahead by the passes) and the
//               Newton point (x, v: 10 doubles per node) written by the roll-out and read back by the stage lanes.
// Instances of one wave advance in lock-step; each carries its own mode (running / least-squares restart pending / done) so that the
// two-attempt logic of k_solve is preserved per instance; a wave ends when its four instances are done.

constexpr int NR_REC = 24;     // node record: Qhat[10] | qhat[8] | Rhat[2] | rhat[2] | 0 | 0   (Rhat, rhat of the stage that LEAVES the node)
constexpr int XR_REC = 10;     // node record: x[8] | v[2]
constexpr int ST_REC = 30;     // LDS stage record: K[2][8] | Sinv00 Sinv01 Sinv11 - | kff[2] | P c [8]
__host__ __device__ inline size_t ws4_len(int N) { return (size_t)(NR_REC + XR_REC) * (N + 1); }
__host__ __device__ inline int lds4_group(int N) { int n = 2 * SB + 96 + ST_REC * N + 8; return n + ((18 - (n & 15)) & 15); }   // == 2 (mod 16): even (16-byte aligned ring chunks), instance regions four banks apart
__host__ __device__ inline size_t lds4_bytes(int N) { return (size_t)(4 * lds4_group(N) + 2 * SB + 64) * sizeof(real); }

PG_DEV real dpp_x1(real v) { return dpp_move<0xB1>(v); }      // value of lane ^ 1 (quad_perm [1,0,3,2])
PG_DEV real g16_sum(real v) {
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) v += __shfl_xor(v, s, 16);
    return v;
}
PG_DEV real g16_max(real v) {
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) { real o = __shfl_xor(v, s, 16); v = o > v ? o : v; }
    return v;
}
PG_DEV real g16_min(real v) {
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) { real o = __shfl_xor(v, s, 16); v = o < v ? o : v; }
    return v;
}

struct StageConst4 { real h0[4], h1[4], bb[NROW], Rd0, wb, wr, wh, Qd6, Qd7, M0, M1; bool act, hji_on; int nrows, s; };

template <int NSLOT, bool PROF>
__global__ __launch_bounds__(64, 1) void k_solve4(DevCfg C, int B, const real* __restrict__ qp, const real* __restrict__ abar, const real* __restrict__ nodes,
                                                  real* __restrict__ ws, SolveOut O, unsigned long long* __restrict__ prof) {
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int slot) { if (PROF) { unsigned long long now = clock64(); pc[slot] += now - tprev; tprev = now; } };
    if (PROF) tprev = clock64();
    const int lane = threadIdx.x, g = lane >> 4, gl = lane & 15, r = gl >> 1, h = gl & 1, j0 = 4 * h;
    const int N = C.N, NN = C.NN;
    int b = blockIdx.x * 4 + g; if (b >= B) b = B - 1;            // a ragged last wave repeats the last instance (identical values to identical addresses)
    extern __shared__ real lds[];
    const int GSZ = lds4_group(N);
    real* const sRing = lds + g * GSZ;            // [2][SB]
    real* const sX = sRing + 2 * SB;              // [8][12] exchange (aliased by the [8][9] transpose buffer)
    real* const sSt = sX + 96;                    // [N][ST_REC]
    real* const sG = sSt + ST_REC * N;            // [8] vector all-gather
    real* const sDum = lds + 4 * GSZ + 2 * SB;    // [64] sink for predicated-off stores (the 2 SB doubles below it absorb the idle ring lanes)
    const QpOff o = qp_offsets(N);
    const real* const Qbase = qp + (size_t)b * C.qp_len; const real* Q = Qbase;
    real* const W = ws + (size_t)b * ws4_len(N);
    real* const nr = W; real* const xr = W + NR_REC * NN;
    const real hf = (real)h, w_lo = r < 6 ? real(1.0) : real(0.0), w6 = r == 6 ? real(1.0) : real(0.0), w7 = r == 7 ? real(1.0) : real(0.0);
    const int rr = r < 6 ? r : 5;

    // ---- stage-block ring: 4 instances x 36 double2 chunks per stage = 144 chunks, three per lane (the third only for lanes < 16) ----
    // Global latency here is ~2.5k cycles while a roll-out / vector-recursion stage computes for ~0.5k: the loads must run SEVERAL stages ahead.
    // The pass loops are therefore unrolled by the prefetch depth D with one register set per in-flight block (the hardware returns loads in
    // order, so each stage only waits for the oldest); a block is written to LDS one stage before it is read (two slots).
    const real2* rsrc[3]; real2* rdst[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        int c = lane + 64 * i; bool ok = c < 4 * SB_CHUNKS; int cq = ok ? c / SB_CHUNKS : 0, off = ok ? c % SB_CHUNKS : 0;
        int bq = blockIdx.x * 4 + cq; if (bq >= B) bq = B - 1;
        rsrc[i] = reinterpret_cast<const real2*>(abar + (size_t)bq * N * SB) + off;
        rdst[i] = ok ? reinterpret_cast<real2*>(lds + cq * GSZ) + off : reinterpret_cast<real2*>(sDum) + (lane & 15) - SB_CHUNKS;   // (- SB_CHUNKS: slot 1 of the sink stays inside it)
    }
    struct RSet { real2 v[3]; };
    auto ring_slot = [&](int k) -> real* { return sRing + (k & 1) * SB; };
    auto rs_load = [&](RSet& S, int k) {
        int kk = k < 0 ? 0 : (k >= N ? N - 1 : k);
#pragma unroll
        for (int i = 0; i < 3; i++) S.v[i] = rsrc[i][kk * SB_CHUNKS];
    };
    auto rs_put = [&](const RSet& S, int k) {
        const int so = (k & 1) * SB_CHUNKS;
#pragma unroll
        for (int i = 0; i < 3; i++) rdst[i][so] = S.v[i];
    };

    const real x0r = r < 6 ? Q[o.qcurr + r] : Q[o.ucurr + r - 6];

    // ---- per-stage constants (re-read from L2 by each stage phase: they would cost 96 VGPRs to keep) ----
    const int n_hji = C.cp.N_HJI < C.Ns ? C.cp.N_HJI : C.Ns;
    auto load_sc = [&](int u) {
        StageConst4 c;
        c.act = gl + 16 * u < N; c.s = c.act ? gl + 16 * u : 0;
        const int s = c.s;
        const real* Q = Qbase; asm volatile("" : "+v"(Q));       // keep the compiler from hoisting these loop-invariant loads into 100+ live registers
        c.hji_on = c.act && (s + 1 < n_hji); c.nrows = c.hji_on ? 16 : 14;
        const real dts = Q[o.dt + s];
        c.Rd0 = real(2.0) * C.cp.R_ddelta / dts; c.wb = C.cp.W_beta * dts; c.wr = C.cp.W_r * dts; c.wh = C.cp.W_HJI;
        c.Qd6 = real(2.0) * C.cp.R_delta * dts; c.Qd7 = real(2.0) * C.cp.R_Fx * dts; c.M0 = Q[o.M]; c.M1 = Q[o.M + 1];
#pragma unroll
        for (int i = 0; i < 4; i++) { c.h0[i] = Q[o.H + 8 * s + 2 * i]; c.h1[i] = Q[o.H + 8 * s + 2 * i + 1]; c.bb[6 + i] = Q[o.G + 4 * s + i]; }
        c.bb[0] = -C.cp.V_min; c.bb[1] = C.cp.V_max; c.bb[2] = -C.fxmin_n; c.bb[3] = Q[o.dmax + s]; c.bb[4] = -Q[o.dmin + s]; c.bb[5] = Q[o.fxmax + s];
        c.bb[10] = real(0.0); c.bb[11] = real(0.0); c.bb[12] = Q[o.ddmax + s]; c.bb[13] = -Q[o.ddmin + s]; c.bb[14] = Q[o.b]; c.bb[15] = real(0.0);
        return c;
    };
    auto slacks = [&](const StageConst4& c, const real* x, real v0, real s1, real s2, real sh, real* out) {
        out[0] = x[1] + c.bb[0]; out[1] = c.bb[1] - x[1]; out[2] = x[7] + c.bb[2]; out[3] = c.bb[3] - x[6]; out[4] = x[6] + c.bb[4]; out[5] = c.bb[5] - x[7];
#pragma unroll
        for (int i = 0; i < 4; i++) out[6 + i] = c.bb[6 + i] - (c.h0[i] * x[2] + c.h1[i] * x[3]) + (i < 2 ? s1 : s2);
        out[10] = s1; out[11] = s2; out[12] = c.bb[12] - v0; out[13] = v0 + c.bb[13];
        out[14] = c.bb[14] + c.M0 * x[6] + c.M1 * x[7] + sh; out[15] = sh;
    };

    // entries of the stage costs that never change, zero padding, node 0
    for (int u = 0; u < NSLOT; u++) {
        const bool act = gl + 16 * u < N; const int s = act ? gl + 16 * u : 0;
        if (act) {
            const real dts = Q[o.dt + s];
            real* n1 = nr + NR_REC * (s + 1);
#pragma unroll
            for (int i = 0; i < 18; i++) n1[i] = real(0.0);
            n1[0] = real(2.0) * C.cp.Q_ds * dts; n1[4] = real(2.0) * C.cp.Q_dpsi * dts; n1[5] = real(2.0) * C.cp.Q_e * dts;
            n1[22] = real(0.0); n1[23] = real(0.0);
            real* n0 = nr + NR_REC * s;
            n0[19] = real(2.0) * C.cp.R_dFx / dts; n0[21] = real(0.0);
            if (s == 0) { for (int i = 0; i < 18; i++) n0[i] = real(0.0); n0[22] = real(0.0); n0[23] = real(0.0); }
            if (s == N - 1) { n1[18] = real(0.0); n1[19] = real(0.0); n1[20] = real(0.0); n1[21] = real(0.0); }
        }
    }
    // slot of Qhat[r][j0+i] inside the packed node record; off-pattern entries read the stored zero at [23]
    int qoff[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = j0 + i;
        qoff[i] = (r == j) ? r : (((r == 2 && j == 3) || (r == 3 && j == 2)) ? 8 : (((r == 6 && j == 7) || (r == 7 && j == 6)) ? 9 : 23));
    }
    __syncthreads();

    // ---- roll-out: lane (r, .) carries x_k[r]; the 8 components are all-gathered through LDS once per stage ----
    constexpr int DF = 6;          // prefetch depth of the short-stage passes (roll-out, vector recursion)
    auto forward = [&](auto use_gain_t) {
        constexpr bool use_gain = decltype(use_gain_t)::value;
        real xi = x0r;
        if (h == 0) xr[r] = xi;
        RSet S[DF];
        { RSet t0; rs_load(t0, 0); rs_put(t0, 0); }
#pragma unroll
        for (int d = 0; d < DF; d++) rs_load(S[d], 1 + d);
        auto body = [&](int k, RSet& Sk) {
            rs_put(Sk, k + 1); rs_load(Sk, k + 1 + DF);
            const real* Rk = ring_slot(k); const real* St = sSt + ST_REC * k; const real* Ar = Rk + SB_ROW * rr;
            sG[r] = xi;
            real A8[8], K0[8], K1[8];
#pragma unroll
            for (int m = 0; m < 8; m++) { A8[m] = Ar[m]; K0[m] = use_gain ? St[m] : real(0.0); K1[m] = use_gain ? St[8 + m] : real(0.0); }
            real kf0 = use_gain ? St[20] : real(0.0), kf1 = use_gain ? St[21] : real(0.0);
            real cr = Rk[SB_C + rr], bf0 = Rk[SB_B + 2 * rr], bf1 = Rk[SB_B + 2 * rr + 1];
            __syncthreads();
            real xm[8];
#pragma unroll
            for (int m = 0; m < 8; m++) xm[m] = sG[m];
            real a0 = kf0, a1 = real(0.0), b0 = kf1, b1 = real(0.0), e0 = cr, e1 = real(0.0);
#pragma unroll
            for (int m = 0; m < 8; m += 2) {
                a0 += K0[m] * xm[m]; a1 += K0[m + 1] * xm[m + 1]; b0 += K1[m] * xm[m]; b1 += K1[m + 1] * xm[m + 1];
                e0 += A8[m] * xm[m]; e1 += A8[m + 1] * xm[m + 1];
            }
            real v0 = a0 + a1, v1 = b0 + b1;
            real xn = w_lo * ((e0 + e1) + (bf0 * v0 + bf1 * v1)) + w6 * (xm[6] + v0) + w7 * (xm[7] + v1);
            xi = xn;
            if (h == 0) xr[XR_REC * (k + 1) + r] = xn;
            if (gl == 0) { xr[XR_REC * k + 8] = v0; xr[XR_REC * k + 9] = v1; }
        };
#pragma unroll 1
        for (int k = 0; k < N; k += DF) {
#pragma unroll
            for (int d = 0; d < DF; d++) if (k + d < N) body(k + d, S[d]);
        }
        __syncthreads();
    };

    // ---- Riccati matrix pass (+ the predictor's vector recursion) ----
    auto riccati_matrices = [&]() {
        const real* nN = nr + NR_REC * N;
        real P4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) P4[i] = nN[qoff[i]];
        real pv = nN[10 + r];
        constexpr int DM = 2;
        struct NRec { real q[4], qv, R0, R1, r0, r1; };
        auto load_rec = [&](NRec& c, int k) {
            const real* n = nr + NR_REC * (k < 0 ? 0 : k);
#pragma unroll
            for (int i = 0; i < 4; i++) c.q[i] = n[qoff[i]];
            c.qv = n[10 + r]; c.R0 = n[18]; c.R1 = n[19]; c.r0 = n[20]; c.r1 = n[21];
        };
        RSet S[DM]; NRec NR[DM];
        { RSet t0; rs_load(t0, N - 1); rs_put(t0, N - 1); }
#pragma unroll
        for (int d = 0; d < DM; d++) { rs_load(S[d], N - 2 - d); load_rec(NR[d], N - 1 - d); }
        auto body = [&](int k, RSet& Sk, NRec& Nk) {
            rs_put(Sk, k - 1); rs_load(Sk, k - 1 - DM);
            real cq[4];
#pragma unroll
            for (int i = 0; i < 4; i++) cq[i] = Nk.q[i];
            const real cqv = Nk.qv, cR0 = Nk.R0, cR1 = Nk.R1, cr0 = Nk.r0, cr1 = Nk.r1;
            load_rec(Nk, k - DM);
            const real* Ak = ring_slot(k); const real* Bk = Ak + SB_B; const real* ck = Ak + SB_C;
            // full row r of P_{k+1}: own half + the partner lane's half
            real Pf[8];
#pragma unroll
            for (int i = 0; i < 4; i++) { real qp_ = dpp_x1(P4[i]); Pf[i] = h ? qp_ : P4[i]; Pf[4 + i] = h ? P4[i] : qp_; }
            // row r of P [Abar | Bbar | cbar], this lane's share of the columns
            real ma[4] = {real(0.0), real(0.0), hf * Pf[6], hf * Pf[7]};                // identity rows 6,7 of Abar touch columns 6,7 only
            real mb = h ? Pf[7] : Pf[6], mc = real(0.0);                            // identity rows of Bbar
#pragma unroll
            for (int m = 0; m < 6; m++) {
#pragma unroll
                for (int i = 0; i < 4; i++) ma[i] += Pf[m] * Ak[SB_ROW * m + j0 + i];
                mb += Pf[m] * Bk[2 * m + h]; mc += Pf[m] * ck[m];
            }
            real* Xr = sX + 12 * r;
#pragma unroll
            for (int i = 0; i < 4; i++) Xr[j0 + i] = ma[i];
            Xr[8 + h] = mb; Xr[10 + h] = h ? pv : mc;
            // stage constants that do not depend on the exchange
            real ar[6], b0[6], b1[6];
#pragma unroll
            for (int m = 0; m < 6; m++) { ar[m] = Ak[SB_ROW * m + r]; b0[m] = Bk[2 * m]; b1[m] = Bk[2 * m + 1]; }
            __syncthreads();
            real MA[8][4], MAr[8], MB0[8], MB1[8], Y[8];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const real* Xm = sX + 12 * m;
#pragma unroll
                for (int i = 0; i < 4; i++) MA[m][i] = Xm[j0 + i];
                MAr[m] = Xm[r]; MB0[m] = Xm[8]; MB1[m] = Xm[9]; Y[m] = Xm[10] + Xm[11];          // y = P c + p_{k+1}
            }
            real F0[4], F1[4], F0r = MAr[6], F1r = MAr[7];
            real S00 = cR0 + MB0[6], S01 = MB1[6], S11 = cR1 + MB1[7];
#pragma unroll
            for (int i = 0; i < 4; i++) { F0[i] = MA[6][i]; F1[i] = MA[7][i]; }
#pragma unroll
            for (int m = 0; m < 6; m++) {
#pragma unroll
                for (int i = 0; i < 4; i++) { F0[i] += b0[m] * MA[m][i]; F1[i] += b1[m] * MA[m][i]; }
                F0r += b0[m] * MAr[m]; F1r += b1[m] * MAr[m];
                S00 += b0[m] * MB0[m]; S01 += b0[m] * MB1[m]; S11 += b1[m] * MB1[m];
            }
            const real idet = frcp(S00 * S11 - S01 * S01);
            const real I00 = S11 * idet, I01 = -S01 * idet, I11 = S00 * idet;
            real K0[4], K1[4], Pn[4];
            const real K0r = -(I00 * F0r + I01 * F1r), K1r = -(I01 * F0r + I11 * F1r);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                K0[i] = -(I00 * F0[i] + I01 * F1[i]); K1[i] = -(I01 * F0[i] + I11 * F1[i]);
                real pn = cq[i] + F0r * K0[i] + F1r * K1[i] + w6 * MA[6][i] + w7 * MA[7][i];
#pragma unroll
                for (int m = 0; m < 6; m++) pn += ar[m] * MA[m][i];
                Pn[i] = pn;
            }
            // vector recursion of the predictor
            real f0 = cr0 + Y[6], f1 = cr1 + Y[7], pvn = cqv + w6 * Y[6] + w7 * Y[7];
#pragma unroll
            for (int m = 0; m < 6; m++) { f0 += b0[m] * Y[m]; f1 += b1[m] * Y[m]; pvn += ar[m] * Y[m]; }
            pvn += K0r * f0 + K1r * f1;
            // per-stage results for the later passes
            real* St = sSt + ST_REC * k;
            {
                real* d = r < 2 ? St + 8 * r + j0 : sDum + 4 * (lane & 15);
#pragma unroll
                for (int i = 0; i < 4; i++) d[i] = r == 0 ? K0[i] : K1[i];
                real* e = gl == 4 ? St + 16 : sDum + 4 * (lane & 15);
                e[0] = I00; e[1] = I01; e[2] = I11;
                real* f = gl == 6 ? St + 20 : sDum + 4 * (lane & 15);
                f[0] = -(I00 * f0 + I01 * f1); f[1] = -(I01 * f0 + I11 * f1);
                *(h == 0 ? St + 22 + r : sDum + lane) = mc;
            }
            // symmetrise through the (now dead) exchange buffer: lane (j, .) holds the transposed entry.  Without it the antisymmetric rounding
            // error of the recursion is amplified by |eig(Abar)|^2 per stage (see k_solve)
            real* Yt = sX;
#pragma unroll
            for (int i = 0; i < 4; i++) Yt[9 * r + j0 + i] = Pn[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; i++) P4[i] = real(0.5) * (Pn[i] + Yt[9 * (j0 + i) + r]);
            pv = pvn;
        };
#pragma unroll 1
        for (int k = N - 1; k >= 0; k -= DM) {
#pragma unroll
            for (int d = 0; d < DM; d++) if (k - d >= 0) body(k - d, S[d], NR[d]);
        }
        __syncthreads();
    };

    // ---- Riccati vector pass of the corrector ----
    auto riccati_vectors = [&]() {
        real pv = nr[NR_REC * N + 10 + r];
        struct VRec { real qv, r0, r1; };
        auto load_rec = [&](VRec& c, int k) { const real* n = nr + NR_REC * (k < 0 ? 0 : k); c.qv = n[10 + r]; c.r0 = n[20]; c.r1 = n[21]; };
        RSet S[DF]; VRec VR[DF];
        { RSet t0; rs_load(t0, N - 1); rs_put(t0, N - 1); }
#pragma unroll
        for (int d = 0; d < DF; d++) { rs_load(S[d], N - 2 - d); load_rec(VR[d], N - 1 - d); }
        auto body = [&](int k, RSet& Sk, VRec& Vk) {
            rs_put(Sk, k - 1); rs_load(Sk, k - 1 - DF);
            const real cqv = Vk.qv, cr0 = Vk.r0, cr1 = Vk.r1;
            load_rec(Vk, k - DF);
            const real* Ak = ring_slot(k); const real* Bk = Ak + SB_B; real* St = sSt + ST_REC * k;
            sG[r] = pv;
            real ar[6], b0[6], b1[6], mcv[8];
#pragma unroll
            for (int m = 0; m < 6; m++) { ar[m] = Ak[SB_ROW * m + r]; b0[m] = Bk[2 * m]; b1[m] = Bk[2 * m + 1]; }
#pragma unroll
            for (int m = 0; m < 8; m++) mcv[m] = St[22 + m];
            const real K0r = St[r], K1r = St[8 + r], I00 = St[16], I01 = St[17], I11 = St[18];
            __syncthreads();
            real Y[8];
#pragma unroll
            for (int m = 0; m < 8; m++) Y[m] = mcv[m] + sG[m];
            real f0 = cr0 + Y[6], f1 = cr1 + Y[7], pvn = cqv + w6 * Y[6] + w7 * Y[7];
#pragma unroll
            for (int m = 0; m < 6; m++) { f0 += b0[m] * Y[m]; f1 += b1[m] * Y[m]; pvn += ar[m] * Y[m]; }
            pvn += K0r * f0 + K1r * f1;
            real* f = gl == 0 ? St + 20 : sDum + 4 * (lane & 15);
            f[0] = -(I00 * f0 + I01 * f1); f[1] = -(I01 * f0 + I11 * f1);
            pv = pvn;
        };
#pragma unroll 1
        for (int k = N - 1; k >= 0; k -= DF) {
#pragma unroll
            for (int d = 0; d < DF; d++) if (k - d >= 0) body(k - d, S[d], VR[d]);
        }
        __syncthreads();
    };

    // ---- stage-parallel state: lane gl owns stages gl + 16 u ----
    // (1/t is recomputed where it is needed and the Newton point is re-read from L2 rather than kept: registers are the scarce resource here)
    real T[NSLOT][NROW], L[NSLOT][NROW], CO[NSLOT][NROW], E[NSLOT][12];
    // assemble the Newton system of one stage at (T, L) with complementarity target sigmu - corr (slack columns eliminated): see k_solve
    auto assemble = [&](int u, const StageConst4& c, real sigmu, bool matrices, bool store) {
        real Wt[NROW], ell[NROW];
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            bool on = j < c.nrows;
            const real itj = frcp(T[u][j]);
            Wt[j] = on ? L[u][j] * itj : real(0.0);
            ell[j] = on ? (sigmu - CO[u][j]) * itj + L[u][j] - Wt[j] * c.bb[j] : real(0.0);
        }
        real g1 = -ell[0] + ell[1], g7 = -ell[2] + ell[5] - c.M1 * ell[14], g6 = ell[3] - ell[4] - c.M0 * ell[14];
        real g2 = real(0.0), g3 = real(0.0);
#pragma unroll
        for (int i = 0; i < 4; i++) { g2 += c.h0[i] * ell[6 + i]; g3 += c.h1[i] * ell[6 + i]; }
        real e_g1 = c.wb - ell[6] - ell[7] - ell[10], e_g2 = c.wr - ell[8] - ell[9] - ell[11], e_gh = c.wh - ell[14] - ell[15];
        real gv0 = ell[12] - ell[13];
        real e_d1 = frcp(Wt[6] + Wt[7] + Wt[10]), e_d2 = frcp(Wt[8] + Wt[9] + Wt[11]), e_dh = c.hji_on ? frcp(Wt[14] + Wt[15]) : real(1.0);
        real e_c10 = -(Wt[6] * c.h0[0] + Wt[7] * c.h0[1]), e_c11 = -(Wt[6] * c.h1[0] + Wt[7] * c.h1[1]);
        real e_c20 = -(Wt[8] * c.h0[2] + Wt[9] * c.h0[3]), e_c21 = -(Wt[8] * c.h1[2] + Wt[9] * c.h1[3]);
        real e_ch0 = Wt[14] * c.M0, e_ch1 = Wt[14] * c.M1;
        if (!c.hji_on) e_gh = real(0.0);
        E[u][0] = e_d1; E[u][1] = e_c10; E[u][2] = e_c11; E[u][3] = e_g1; E[u][4] = e_d2; E[u][5] = e_c20; E[u][6] = e_c21; E[u][7] = e_g2;
        E[u][8] = e_dh; E[u][9] = e_ch0; E[u][10] = e_ch1; E[u][11] = e_gh;
        if (c.act && store) {
            real* n1 = nr + NR_REC * (c.s + 1); real* qo = n1 + 10; real* n0 = nr + NR_REC * c.s;
            qo[1] = g1;
            qo[2] = g2 - e_c10 * e_g1 * e_d1 - e_c20 * e_g2 * e_d2;
            qo[3] = g3 - e_c11 * e_g1 * e_d1 - e_c21 * e_g2 * e_d2;
            qo[6] = g6 - e_ch0 * e_gh * e_dh; qo[7] = g7 - e_ch1 * e_gh * e_dh;
            n0[20] = gv0;
            if (matrices) {
                n1[1] = Wt[0] + Wt[1];
                n1[6] = c.Qd6 + Wt[3] + Wt[4] + c.M0 * c.M0 * Wt[14] - e_ch0 * e_ch0 * e_dh;
                n1[7] = c.Qd7 + Wt[2] + Wt[5] + c.M1 * c.M1 * Wt[14] - e_ch1 * e_ch1 * e_dh;
                real yy = real(0.0), yr = real(0.0), rr_ = real(0.0);
#pragma unroll
                for (int i = 0; i < 4; i++) { yy += Wt[6 + i] * c.h0[i] * c.h0[i]; yr += Wt[6 + i] * c.h0[i] * c.h1[i]; rr_ += Wt[6 + i] * c.h1[i] * c.h1[i]; }
                n1[2] = yy - e_c10 * e_c10 * e_d1 - e_c20 * e_c20 * e_d2;
                n1[8] = yr - e_c10 * e_c11 * e_d1 - e_c20 * e_c21 * e_d2;
                n1[3] = rr_ - e_c11 * e_c11 * e_d1 - e_c21 * e_c21 * e_d2;
                n1[9] = c.M0 * c.M1 * Wt[14] - e_ch0 * e_ch1 * e_dh;
                n0[18] = c.Rd0 + Wt[12] + Wt[13];
            }
        }
    };
    // Newton point of the lane's stage (x, v from the roll-out; eliminated slacks recovered) and the slack of every row there
    struct NP { real xn[8], sn1, sn2, snh; };
    auto newton_point = [&](int u, const StageConst4& c, NP& p, real* tplus) {
        const real* xk = xr + XR_REC * (c.s + 1);
#pragma unroll
        for (int m = 0; m < 8; m++) p.xn[m] = xk[m];
        const real vn0 = xr[XR_REC * c.s + 8];
        p.sn1 = -(E[u][1] * p.xn[2] + E[u][2] * p.xn[3] + E[u][3]) * E[u][0];
        p.sn2 = -(E[u][5] * p.xn[2] + E[u][6] * p.xn[3] + E[u][7]) * E[u][4];
        p.snh = c.hji_on ? -(E[u][9] * p.xn[6] + E[u][10] * p.xn[7] + E[u][11]) * E[u][8] : real(0.0);
        slacks(c, p.xn, vn0, p.sn1, p.sn2, p.snh, tplus);
    };
    auto SXp = [&](const StageConst4& c) -> real* { return O.sol_x + (size_t)b * NN * 8 + 8 * (c.s + 1); };
    auto SGp = [&](const StageConst4& c) -> real* { return O.sol_sigma + ((size_t)b * N + c.s) * 3; };

    // ---- first start: v = 0 roll-out (dynamics- and rate-feasible), sigma just feasible; t = max(slack, tau); lambda = mu0 / t ----
    forward(std::false_type{});
    stamp(4);
    real rp0 = real(0.0), ntot = real(0.0);
#pragma unroll
    for (int u = 0; u < NSLOT; u++) {
        const StageConst4 c = load_sc(u);
        real xs[8];
#pragma unroll
        for (int m = 0; m < 8; m++) xs[m] = xr[XR_REC * (c.s + 1) + m];
        real sl[NROW];
        slacks(c, xs, real(0.0), real(0.0), real(0.0), real(0.0), sl);
        const real sig0 = real(0.1), tau = real(1e-4);
        real sg1 = fmax(real(0.0), -fmin(sl[6], sl[7])) + sig0, sg2 = fmax(real(0.0), -fmin(sl[8], sl[9])) + sig0, sgh = c.hji_on ? fmax(real(0.0), -sl[14]) + sig0 : real(0.0);
        slacks(c, xs, real(0.0), sg1, sg2, sgh, sl);
        if (c.act) {
            real* SX = SXp(c); real* SG = SGp(c);
#pragma unroll
            for (int m = 0; m < 8; m++) SX[m] = xs[m];
            SG[0] = sg1; SG[1] = sg2; SG[2] = sgh;
        }
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            bool on = c.act && j < c.nrows;
            real tj = on ? fmax(sl[j], tau) : real(1.0);
            T[u][j] = tj; L[u][j] = on ? C.ipm_mu0 / tj : real(0.0); CO[u][j] = real(0.0);
            if (on) rp0 = fmax(rp0, tj - sl[j]);
        }
        ntot += c.act ? (real)c.nrows : real(0.0);
    }
    rp0 = g16_max(rp0); ntot = g16_sum(ntot);

    int non[NSLOT];                       // rows that exist in the lane's stage u (0 for an inactive lane)
#pragma unroll
    for (int u = 0; u < NSLOT; u++) { const bool act = gl + 16 * u < N; non[u] = act ? ((gl + 16 * u + 1 < n_hji) ? 16 : 14) : 0; }
    enum { RUN = 0, LSQ = 1, DONE = 2 };
    int mode = RUN, attempt = 0, it = 0, it_total = 0, status = PG_MAX_ITER;
    real phi = real(1.0), mu = real(0.0);
    bool fail = false;
    const int guard_max = 4 * C.ipm_max_iter + 8;
#pragma unroll 1
    for (int guard = 0; guard < guard_max; guard++) {
        // ---- top of the iteration: duality measure, termination, restart decision (per instance) ----
        real musum = real(0.0);
#pragma unroll
        for (int u = 0; u < NSLOT; u++)
#pragma unroll
            for (int j = 0; j < NROW; j++) musum += j < non[u] ? T[u][j] * L[u][j] : real(0.0);
        const real mu_now = g16_sum(musum) / ntot;
        if (mode == RUN) {
            mu = mu_now;
            const int cap = attempt == 0 ? C.ipm_max_iter : 3 * C.ipm_max_iter;
            if (!(mu == mu) || fabs(mu) > PG_BIG) { status = PG_NUMERICAL; mode = DONE; }
            else if (mu <= C.ipm_tol && phi * fmax(rp0, real(1.0)) <= C.ipm_tol) { status = PG_SOLVED; mode = DONE; }
            else if (fail || it >= cap) {
                if (attempt == 0) { it_total += it; it = 0; attempt = 1; mode = LSQ; fail = false; }
                else { status = PG_MAX_ITER; mode = DONE; }
            }
        }
        if (!__any(mode != DONE)) break;

        // ---- predictor (sigma = 0, no correction) or the least-squares start: one Newton system with fresh matrices ----
#pragma unroll
        for (int u = 0; u < NSLOT; u++) {
            const StageConst4 c = load_sc(u);
            if (mode == LSQ) {
#pragma unroll
                for (int j = 0; j < NROW; j++) { T[u][j] = real(1.0); L[u][j] = (c.act && j < c.nrows) ? real(1.0) : real(0.0); CO[u][j] = real(0.0); }
            } else if (mode == RUN) {
#pragma unroll
                for (int j = 0; j < NROW; j++) CO[u][j] = real(0.0);
            }
            assemble(u, c, real(0.0), true, mode != DONE);
        }
        __syncthreads();
        stamp(0);
        riccati_matrices();
        stamp(1);
        forward(std::true_type{});
        stamp(3);

        // the slack of every row at the predictor's Newton point is parked in CO (free until the corrector's second-order term is formed)
        real rmax = real(0.0), tmin = PG_BIG, msum_a = real(0.0);
#pragma unroll
        for (int u = 0; u < NSLOT; u++) {
            const StageConst4 c = load_sc(u);
            NP np;
            newton_point(u, c, np, CO[u]);
#pragma unroll
            for (int j = 0; j < NROW; j++) {
                bool on = c.act && j < c.nrows;
                const real itj = frcp(T[u][j]);
                real dt_ = CO[u][j] - T[u][j], dl_ = -(L[u][j] * itj) * CO[u][j];
                real rj = fmax(-dt_ * itj, -dl_ * frcp(L[u][j]));
                rmax = fmax(rmax, on ? rj : real(0.0));
                tmin = fmin(tmin, on ? CO[u][j] : PG_BIG);
            }
        }
        rmax = g16_max(rmax); tmin = g16_min(tmin);
        const real aaff = rmax > real(1.0) ? real(1.0) / rmax : real(1.0);
        real sg = real(0.0);
        bool doB = false, restarted = false;
        if (mode == LSQ) {
            // uniform shift that makes every slack of the least-squares point >= 1
            const real shift = tmin < real(1.0) ? real(1.0) - tmin : real(0.0);
#pragma unroll
            for (int u = 0; u < NSLOT; u++) {
                const StageConst4 c = load_sc(u);
                NP np; real tp[NROW];
                newton_point(u, c, np, tp);
                if (c.act) {
                    real* SX = SXp(c); real* SG = SGp(c);
#pragma unroll
                    for (int m = 0; m < 8; m++) SX[m] = np.xn[m];
                    SG[0] = np.sn1; SG[1] = np.sn2; SG[2] = np.snh;
                }
#pragma unroll
                for (int j = 0; j < NROW; j++) {
                    bool on = c.act && j < c.nrows;
                    real tj = on ? tp[j] + shift : real(1.0);
                    T[u][j] = tj; L[u][j] = on ? C.ipm_mu0 / tj : real(0.0); CO[u][j] = real(0.0);
                }
            }
            rp0 = shift; phi = real(1.0); mode = RUN; restarted = true;
        } else {
            // second-order term dt * dl of the affine direction replaces the parked slack; mu after the affine step
#pragma unroll
            for (int u = 0; u < NSLOT; u++)
#pragma unroll
                for (int j = 0; j < NROW; j++) {
                    const real itj = frcp(T[u][j]);
                    real dt_ = CO[u][j] - T[u][j], dl_ = -(L[u][j] * itj) * CO[u][j];
                    msum_a += j < non[u] ? (T[u][j] + aaff * dt_) * (L[u][j] + aaff * dl_) : real(0.0);
                    CO[u][j] = dt_ * dl_;
                }
        }
        const real mu_aff = g16_sum(msum_a) / ntot;
        if (mode == RUN && !restarted) {
            // rounding floor (see k_solve): mu within 1e4x of the tolerance and the affine direction cannot move any more
            if (mu <= real(1e4) * C.ipm_tol && aaff < real(0.3) && phi * fmax(rp0, real(1.0)) <= C.ipm_tol) { status = PG_SOLVED; mode = DONE; }
            else { sg = fmin(mu_aff / mu, real(1.0)); sg = sg * sg * sg; doB = true; }
        }

        // ---- corrector ----
        if (__any(doB)) {
#pragma unroll
            for (int u = 0; u < NSLOT; u++) {
                const StageConst4 c = load_sc(u);
                assemble(u, c, sg * mu, false, doB);
            }
            __syncthreads();
            stamp(0);
            riccati_vectors();
            stamp(2);
            forward(std::true_type{});
            stamp(3);
            // the corrector's direction is formed twice from the re-read Newton point (step length, then update) instead of being kept in registers
            real rmx = real(0.0);
#pragma unroll
            for (int u = 0; u < NSLOT; u++) {
                const StageConst4 c = load_sc(u);
                NP np; real tp[NROW];
                newton_point(u, c, np, tp);
#pragma unroll
                for (int j = 0; j < NROW; j++) {
                    bool on = c.act && j < c.nrows;
                    const real itj = frcp(T[u][j]);
                    real dt_ = tp[j] - T[u][j], dl_ = (sg * mu - CO[u][j]) * itj - (L[u][j] * itj) * tp[j];
                    real rj = fmax(-dt_ * itj, -dl_ * frcp(L[u][j]));
                    rmx = fmax(rmx, on ? rj : real(0.0));
                }
            }
            rmx = g16_max(rmx);
            if (doB) {
                const real alpha = rmx > real(0.995) ? real(0.995) / rmx : real(1.0);
#pragma unroll
                for (int u = 0; u < NSLOT; u++) {
                    const StageConst4 c = load_sc(u);
                    NP np; real tp[NROW];
                    newton_point(u, c, np, tp);
#pragma unroll
                    for (int j = 0; j < NROW; j++) {
                        bool on = c.act && j < c.nrows;
                        const real itj = frcp(T[u][j]);
                        real dt_ = tp[j] - T[u][j], dl_ = (sg * mu - CO[u][j]) * itj - (L[u][j] * itj) * tp[j];
                        T[u][j] += on ? alpha * dt_ : real(0.0); L[u][j] += on ? alpha * dl_ : real(0.0);
                    }
                    if (c.act) {
                        real* SX = SXp(c); real* SG = SGp(c);
#pragma unroll
                        for (int m = 0; m < 8; m++) { real cx = SX[m]; SX[m] = cx + alpha * (np.xn[m] - cx); }
                        real c1 = SG[0], c2 = SG[1], c3 = SG[2];
                        SG[0] = c1 + alpha * (np.sn1 - c1); SG[1] = c2 + alpha * (np.sn2 - c2); SG[2] = c3 + alpha * (np.snh - c3);
                    }
                }
                phi *= (real(1.0) - alpha);
                it++;
                if (mu > real(1e8) * C.ipm_mu0) fail = true;          // diverging: give up on this start
            }
        }
    }
    stamp(0);
    if (PROF && gl == 0) { for (int i = 0; i < 6; i++) prof[(size_t)b * 6 + i] = pc[i]; }
    it_total += it;
    if (mode != DONE) status = PG_MAX_ITER;
    if (status == PG_SOLVED) {
        const real Ux0 = Q[o.qcurr + 1], Fx0 = Q[o.ucurr + 1];
        if (Ux0 < C.cp.V_min || Ux0 > C.cp.V_max || Fx0 < C.fxmin_n) status = PG_INFEASIBLE_X0;
    }
    // ---- outputs ----
    __syncthreads();
    real* SX0 = O.sol_x + (size_t)b * NN * 8;
    if (h == 0) SX0[r] = x0r;
#pragma unroll
    for (int u = 0; u < NSLOT; u++) {
        const bool act = gl + 16 * u < N; const int s = act ? gl + 16 * u : 0;
        const int nrows = (act && (s + 1 < n_hji)) ? 16 : 14;
        if (act) {
            unsigned mask = 0;
#pragma unroll
            for (int j = 0; j < NROW; j++) if (j < nrows && L[u][j] > T[u][j]) mask |= (1u << j);
            O.active[(size_t)b * N + s] = (uint16_t)mask;
        }
    }
    if (gl == 0) {
        // get_next_control: coupled_lat_long.jl:370-374 (node 2 of the reference = stage 0's node)
        const real* S1 = O.sol_x + (size_t)b * NN * 8 + 8;
        real d = S1[6] * C.un0, Fx = S1[7] * C.un1;
        if (C.formulation == PG_DECOUPLED) Fx = nodes[((size_t)b * NN + 1) * 10 + 7];      // decoupled_lat_long.jl:275-278
        real* U = O.u_out + (size_t)b * 3;
        U[0] = d; U[1] = Fx > real(0.0) ? Fx * C.veh.fwd_frac : Fx * C.veh.fwb_frac; U[2] = Fx > real(0.0) ? Fx * C.veh.rwd_frac : Fx * C.veh.rwb_frac;
        O.status[b] = status; O.iters[b] = it_total; O.mu[b] = mu; O.solved[b] = 1;          // model_predictive_control.jl:76: solved = true
    }
}
