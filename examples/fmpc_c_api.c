/* The FMPC C-ABI from plain C99 (no C++ on the caller's side): 4 Van der Pol instances (the reference's FmpcProblemOscillator,
   nmpc_fmpc/tests/src/TestFmpcOscillator.cpp:18-135), Variable::reset(0, 0, 0, 1, 1), two warm-started solves; prints status,
   iteration count, last KKT error and the first input of every instance.  Build + run:
     gcc -std=c99 -O2 -Iinclude examples/fmpc_c_api.c -Lnmpc_amd/lib -lnmpc_hip_ddp -Wl,-rpath,$PWD/nmpc_amd/lib -o /tmp/fmpc_c_api
     /tmp/fmpc_c_api                                                                                                        */
#include <stdio.h>
#include <stdlib.h>

#include <nmpc_hip_fmpc.h>

#define B 4
#define T 100
#define MAX_ITER 3

static void check(int rc, const char * what)
{
  if(rc != NMPC_HIP_OK)
  {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, nmpc_hip_fmpc_last_error());
    exit(1);
  }
}

int main(void)
{
  nmpc_hip_fmpc_handle h = NULL;
  nmpc_hip_fmpc_config cfg;
  double t[B] = {0, 0, 0, 0};
  double x0[B][2];
  static double U[B][T][1];
  static double trace[B][MAX_ITER][NMPC_HIP_FMPC_NTRACE];
  int status[B], iters[B];
  int n = 0, m = 0, g = 0, b, pass;
  size_t param_bytes = 0;

  check(nmpc_hip_fmpc_model_info("fmpc_oscillator", &n, &m, &g, &param_bytes), "model_info");
  for(b = 0; b < B; b++)
  {
    x0[b][0] = 0.1 * b; /* instance 0: the reference's initial state (TestFmpcOscillator.cpp:160) */
    x0[b][1] = 1.0;
  }
  check(nmpc_hip_fmpc_create("fmpc_oscillator", T, B, 0, &h), "create");
  check(nmpc_hip_fmpc_default_config(&cfg), "default_config");
  cfg.horizon_steps = T;
  cfg.max_iter = MAX_ITER;
  check(nmpc_hip_fmpc_set_config(h, &cfg), "set_config");
  check(nmpc_hip_fmpc_reset_variable(h, 0.0, 0.0, 0.0, 1.0, 1.0), "reset_variable");
  for(pass = 0; pass < 2; pass++) /* the second solve continues from the resident variable */
  {
    check(nmpc_hip_fmpc_solve(h, t, &x0[0][0]), "solve");
    check(nmpc_hip_fmpc_get(h, NMPC_HIP_FMPC_FIELD_U, U, sizeof(U), 0), "get U");
    check(nmpc_hip_fmpc_get(h, NMPC_HIP_FMPC_FIELD_STATUS, status, sizeof(status), 0), "get status");
    check(nmpc_hip_fmpc_get(h, NMPC_HIP_FMPC_FIELD_ITERS, iters, sizeof(iters), 0), "get iters");
    check(nmpc_hip_fmpc_get(h, NMPC_HIP_FMPC_FIELD_TRACE, trace, sizeof(trace), 0), "get trace");
    for(b = 0; b < B; b++)
    {
      printf("solve %d instance %d dims %d %d %d status %d iter %d kkt_error %.12e u0 %.12e\n", pass, b, n, m, g, status[b], iters[b],
             trace[b][iters[b] - 1][NMPC_HIP_FMPC_TRACE_KKT_ERROR], U[b][0][0]);
    }
  }
  check(nmpc_hip_fmpc_destroy(h), "destroy");
  return 0;
}
