/* Pure-C consumer of include/pigeon_mpc.h: what a foreign host -- the Julia ccall layer of INTEGRATION.md, or any C program --
 * goes through, with no Python and no ctypes in between.  Built with plain gcc against the header, linked to libpigeon_hip.so, run by
 * tests/test_gpu_abi_smoke.py on the GPU box:   create -> set trajectory -> pg_step (B = 2, cold) -> pg_step (warm) -> read-backs -> destroy.
 * Exit status 0 and a line "abi_smoke ok ..." on success; any failed check prints the reason and exits non-zero. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pigeon_mpc.h"

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "abi_smoke FAILED: " __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

int main(void) {
    pg_config cfg;
    int32_t lay[32];
    int nlay = pg_abi_layout(lay, 32);
    CHECK(nlay >= 3 && lay[0] == (int32_t)sizeof(pg_config) && lay[1] == (int32_t)sizeof(pg_vehicle) && lay[2] == (int32_t)sizeof(pg_control_params),
          "struct sizes of this translation unit differ from the library's (%d %d %d vs %zu %zu %zu)", lay[0], lay[1], lay[2], sizeof(pg_config), sizeof(pg_vehicle),
          sizeof(pg_control_params));
    CHECK(pg_default_config(&cfg) == PG_OK, "pg_default_config");
    CHECK(cfg.N_short == 10 && cfg.N_long == 20 && fabs(cfg.vehicle.m - 1964.0) < 1e-9 && cfg.polish == 1, "defaults are not X1 / the reference's keyword values");
    cfg.batch_capacity = 2;
    pg_handle* h = NULL;
    int rc = pg_create(&cfg, &h);
    CHECK(rc == PG_OK && h, "pg_create: %d (%s)", rc, pg_last_error(NULL));

    /* straight_trajectory(200 m, 5 m/s) of trajectories.jl:96-105, sampled every metre (psi measured from North: heading along +N) */
    enum { L = 201 };
    static double t[L], s[L], V[L], A[L], E[L], N[L], psi[L], kap[L];
    for (int i = 0; i < L; i++) { s[i] = i; t[i] = i / 5.0; V[i] = 5.0; A[i] = 0; E[i] = 0; N[i] = i; psi[i] = 0; kap[i] = 0; }
    double u[6]; int32_t st[2], it[2];
    double state[12] = {0.3, 20.0, 0.02, 5.0, 0.0, 0.0, /**/ -0.4, 60.0, -0.03, 4.5, 0.1, 0.01};       /* (E, N, psi, Ux, Uy, r) x 2 */
    double control[6] = {0, 0, 0, /**/ 0.01, 0, 100.0};
    double t0[2] = {4.0, 12.0}, toff[2] = {0.0, 0.0};
    rc = pg_step(h, 2, state, control, t0, NULL, toff, u, st, it);
    CHECK(rc == PG_ERR_STATE, "pg_step without a trajectory must fail with PG_ERR_STATE, got %d", rc);
    CHECK(strlen(pg_last_error(h)) > 0, "pg_last_error is empty after a failure");
    rc = pg_set_trajectory(h, L, t, s, V, A, E, N, psi, kap, NULL, NULL, NULL, NULL);
    CHECK(rc == PG_OK, "pg_set_trajectory: %d (%s)", rc, pg_last_error(h));
    rc = pg_step(h, 2, state, control, t0, NULL, toff, u, st, it);
    CHECK(rc == PG_OK, "pg_step: %d (%s)", rc, pg_last_error(h));
    for (int b = 0; b < 2; b++) {
        CHECK(st[b] == PG_SOLVED, "instance %d: status %d", b, st[b]);
        CHECK(it[b] >= 0 && it[b] <= 40, "instance %d: %d iterations", b, it[b]);      /* 0: served by the active-set guess alone (pg_config.cold_guess) */
        CHECK(isfinite(u[3 * b]) && fabs(u[3 * b]) <= cfg.vehicle.delta_max + 1e-9, "instance %d: steering %g outside the actuator range", b, u[3 * b]);
    }
    double sep[6], ts[2 * 31], x[2 * 31 * 8];
    CHECK(pg_get_path_coordinates(h, sep) == PG_OK && fabs(sep[0] - 20.0) < 1e-5 && fabs(fabs(sep[1]) - 0.3) < 1e-5, "path_coordinates (s, e) = (%g, %g)", sep[0], sep[1]);
    CHECK(pg_get_time_steps(h, ts, NULL, NULL) == PG_OK && ts[0] == 4.0 && fabs(ts[10] - 4.1) < 1e-12 && fabs(ts[11] - 4.2) < 1e-12 && fabs(ts[12] - 4.4) < 1e-12, "time grid %g %g %g", ts[0], ts[10], ts[11]);
    CHECK(pg_get_solution(h, x, NULL) == PG_OK && x[1] == 5.0 && x[8 + 7] * cfg.vehicle.Fx_max != 0.0, "solution read-back");
    int32_t pol[2];
    CHECK(pg_get_polish_info(h, pol) == PG_OK && pol[0] >= 1 && pol[1] >= 1, "polish info %d %d", pol[0], pol[1]);
    /* second step 10 ms later: the warm branch (solved = true inside the handle) */
    double u2[6]; t0[0] += 0.01; t0[1] += 0.01;
    rc = pg_step(h, 2, state, u, t0, NULL, toff, u2, st, it);
    CHECK(rc == PG_OK && PG_IS_SOLVED(st[0]) && PG_IS_SOLVED(st[1]), "warm pg_step: rc %d status %d %d", rc, st[0], st[1]);
    CHECK(fabs(u2[0] - u[0]) < 0.05, "warm step jumps: %g -> %g", u[0], u2[0]);
    CHECK(pg_step(h, 3, state, control, t0, NULL, toff, u, st, it) == PG_ERR_INVALID, "B > batch_capacity must be rejected");
    CHECK(pg_destroy(h) == PG_OK, "pg_destroy");
    printf("abi_smoke ok: %d-bit library, u = (%.6f, %.1f, %.1f) (%.6f, %.1f, %.1f), iterations %d %d\n", pg_precision_bits(), u2[0], u2[1], u2[2], u2[3], u2[4], u2[5], it[0], it[1]);
    return 0;
}
