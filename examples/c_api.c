/* The C-ABI from plain C99 (no C++ anywhere on the caller's side): create a handle for 4 cart-pole instances, solve,
   read the first input and the status of every instance.  Build + run:
     gcc -std=c99 -O2 -Iinclude examples/c_api.c -Lnmpc_amd/lib -lnmpc_hip_ddp -Wl,-rpath,$PWD/nmpc_amd/lib -lm -o /tmp/c_api
     /tmp/c_api                                                                                                        */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <nmpc_hip_ddp.h>

#define B 4
#define T 100

static void check(int rc, const char * what)
{
  if(rc != NMPC_HIP_OK)
  {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, nmpc_hip_ddp_last_error());
    exit(1);
  }
}

int main(void)
{
  nmpc_hip_ddp_handle h = NULL;
  nmpc_hip_ddp_config cfg;
  double t0[B] = {0, 0, 0, 0};
  double x0[B][4];
  static double u_init[B][T][1];
  static double U[B][T][1];
  int status[B], iters[B];
  const char * kernel = NULL;
  int b;

  for(b = 0; b < B; b++)
  {
    x0[b][0] = 0.0;
    x0[b][1] = M_PI - 0.3 * b; /* instance 0: the reference's swing-up start (TestDDPCartPole.cpp:308) */
    x0[b][2] = 0.0;
    x0[b][3] = 0.0;
  }
  check(nmpc_hip_ddp_create("cartpole", T, B, 0, &h), "create");
  check(nmpc_hip_ddp_default_config(&cfg), "default_config");
  cfg.horizon_steps = T;
  check(nmpc_hip_ddp_set_config(h, &cfg), "set_config");
  check(nmpc_hip_ddp_solve(h, t0, &x0[0][0], &u_init[0][0][0]), "solve");
  check(nmpc_hip_ddp_get(h, NMPC_HIP_FIELD_U, U, sizeof(U)), "get U");
  check(nmpc_hip_ddp_get(h, NMPC_HIP_FIELD_STATUS, status, sizeof(status)), "get status");
  check(nmpc_hip_ddp_get(h, NMPC_HIP_FIELD_ITERS, iters, sizeof(iters)), "get iters");
  check(nmpc_hip_ddp_kernel_name(h, &kernel), "kernel_name");
  for(b = 0; b < B; b++)
  {
    printf("instance %d status %d iter %d u0 %.12e (%s)\n", b, status[b], iters[b], U[b][0][0], kernel);
  }
  check(nmpc_hip_ddp_destroy(h), "destroy");
  return 0;
}
