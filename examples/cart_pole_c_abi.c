/* The C ABI of include/mpcqp.h without any Python: the closed loop of pyMPC's examples/example_inverted_pendulum.py:10-69 (cart-pole
 * linearised around the upright position, Ts = 50 ms, Np = 20) -- setup(), then per step output() / plant / update() as one call
 * (mpcqp_mpc_step = MPCController.__controller_function__, mpc.py:377-384).
 *
 *   gcc -O2 -Iinclude examples/cart_pole_c_abi.c -o cart_pole_c_abi -Lpympc_amd -lmpcqp_hip -Wl,-rpath,$PWD/pympc_amd -lm
 *   ./cart_pole_c_abi 40          (the same program links against oracle/libmpcqp_cpu.so -- the CPU twin of the ABI -- with -Loracle -lmpcqp_cpu)
 *
 * Prints one line per step: k, the applied input, the cart position and the pole angle, the solver's iteration count and status.
 * tests/test_c_example.py builds it, runs it on the GPU and compares every printed input with the Python drop-in class. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mpcqp.h"

#define NX 4
#define NU 1
#define NP 20

#define CHECK(call) do { int rc_ = (call); if (rc_ != MPCQP_OK) { fprintf(stderr, "%s: %d (%s)\n", #call, rc_, mpcqp_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    const int nsteps = argc > 1 ? atoi(argv[1]) : 40;
    /* model of the example: M = 0.5, m = 0.2, b = 0.1, ftheta = 0.1, l = 0.3, g = 9.81; forward-Euler discretisation */
    const double M = 0.5, m = 0.2, b = 0.1, ftheta = 0.1, l = 0.3, g = 9.81, Ts = 50e-3;
    const double Ac[NX * NX] = {0.0, 1.0, 0.0, 0.0,
                                0.0, -b / M, -(g * m) / M, (ftheta * m) / M,
                                0.0, 0.0, 0.0, 1.0,
                                0.0, b / (M * l), (M * g + g * m) / (M * l), -(M * ftheta + ftheta * m) / (M * l)};
    const double Bc[NX * NU] = {0.0, 1.0 / M, 0.0, -1.0 / (M * l)};
    double Ad[NX * NX], Bd[NX * NU];
    for (int i = 0; i < NX; ++i) { for (int j = 0; j < NX; ++j) Ad[i * NX + j] = (i == j ? 1.0 : 0.0) + Ts * Ac[i * NX + j]; Bd[i] = Ts * Bc[i]; }
    double Qx[NX * NX] = {0}, QxN[NX * NX] = {0};
    const double qd[NX] = {0.3, 0.0, 1.0, 0.0};
    for (int i = 0; i < NX; ++i) { Qx[i * NX + i] = qd[i]; QxN[i * NX + i] = qd[i]; }
    const double Qu[1] = {0.0}, QDu[1] = {0.01};
    const double xmin[NX] = {-1.0, -100.0, -100.0, -100.0}, xmax[NX] = {0.3, 100.0, 100.0, 100.0};
    const double umin[1] = {-20.0}, umax[1] = {20.0}, Dumin[1] = {-5.0}, Dumax[1] = {5.0};
    const double uref[1] = {0.0}, eps_feas[1] = {1e3};
    double x[NX] = {0.0, 0.0, 15.0 * 2.0 * M_PI / 360.0, 0.0}, um1[1] = {0.0};
    const double xref[NX] = {0.3, 0.0, 0.0, 0.0};

    mpcqp_settings st; mpcqp_default_settings(&st);       /* OSQP's defaults as pyMPC uses them (eps 1e-3, warm start) */
    mpcqp_handle *h = NULL;
    CHECK(mpcqp_create(&h, 0, 1, NX, NU, NP, NP, &st));
    mpcqp_model mdl; memset(&mdl, 0, sizeof(mdl));
    mdl.Ad = Ad; mdl.Bd = Bd; mdl.Qx = Qx; mdl.QxN = QxN; mdl.Qu = Qu; mdl.QDu = QDu;
    mdl.xmin = xmin; mdl.xmax = xmax; mdl.umin = umin; mdl.umax = umax; mdl.Dumin = Dumin; mdl.Dumax = Dumax;
    mdl.uref = uref; mdl.eps_feas = eps_feas;
    CHECK(mpcqp_setup(h, &mdl, x, um1, xref, 1));
    CHECK(mpcqp_solve(h));                                 /* setup(solve=True), mpc.py:254-269 */
    double u[1]; mpcqp_info info;
    CHECK(mpcqp_get_u0(h, u));
    CHECK(mpcqp_get_solution(h, NULL, NULL, &info));
    for (int k = 0; k < nsteps; ++k) {
        printf("%3d u % .15e  p % .6f  theta % .6f  iters %d  %s\n", k, u[0], x[0], x[2], info.iter, mpcqp_status_string(info.status));
        double xn[NX];
        for (int i = 0; i < NX; ++i) { double a = Bd[i] * u[0]; for (int j = 0; j < NX; ++j) a += Ad[i * NX + j] * x[j]; xn[i] = a; }
        memcpy(x, xn, sizeof(x));
        CHECK(mpcqp_mpc_step(h, x, u, NULL, 0, u));        /* update(x, u) + solve + output() */
        CHECK(mpcqp_get_solution(h, NULL, NULL, &info));
    }
    mpcqp_destroy(h);
    return 0;
}
