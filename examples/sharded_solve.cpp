// DDPSolverSharded: one host process, a batch sharded over several devices (or, on a one-GPU box, over several handles of the
// same device), one gather at the end.  Prints whether the gathered results equal the unsharded solve bit for bit.
//   g++ -std=c++17 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/sharded_solve.cpp -Lnmpc_amd/lib -lnmpc_hip_ddp
//       -L/opt/rocm/lib -lamdhip64 [-DNMPC_AMD_WITH_RCCL -lrccl] -Wl,-rpath,$PWD/nmpc_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/sharded
//   /tmp/sharded <model> <batch> <horizon> <shards> [rccl]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <nmpc_amd/DDPSolverSharded.hpp>

int main(int argc, char ** argv)
{
  const std::string model = argc > 1 ? argv[1] : "cartpole";
  const int B = argc > 2 ? std::atoi(argv[2]) : 203, T = argc > 3 ? std::atoi(argv[3]) : 40, S = argc > 4 ? std::atoi(argv[4]) : 2;
  const bool rccl = argc > 5 && std::strcmp(argv[5], "rccl") == 0;
  int n = 0, m = 0, dyn = 0;
  size_t pb = 0;
  if(nmpc_hip_ddp_model_info(model.c_str(), &n, &m, &dyn, &pb) != NMPC_HIP_OK)
  {
    std::printf("unknown model\n");
    return 2;
  }
  const int mm = m > 0 ? m : 1;
  int n_dev = 0;
  hipGetDeviceCount(&n_dev);
  std::vector<int> devices(S);
  for(int s = 0; s < S; s++)
  {
    devices[s] = rccl ? s : s % n_dev; // Gather::Copy may put several shards on one device
  }
  // deterministic inputs
  std::vector<double> x0(static_cast<size_t>(B) * n), u0(static_cast<size_t>(B) * T * mm, 0.0), t0(B, 0.0);
  unsigned long long st = 99;
  for(double & v : x0)
  {
    st = st * 6364136223846793005ULL + 1442695040888963407ULL;
    v = ((st >> 11) * (1.0 / 9007199254740992.0)) * 2 - 1;
  }
  if(model.rfind("quadrotor", 0) == 0)
  {
    for(double & v : x0) v *= 0.3;
    for(double & v : u0) v = 9.80665 / 4;
  }
  try
  {
    nmpc_amd::DDPSolverSharded sharded(model, T, B, devices, rccl ? nmpc_amd::DDPSolverSharded::Gather::Rccl : nmpc_amd::DDPSolverSharded::Gather::Copy);
    sharded.config().max_iter = 5;
    sharded.solve(t0.data(), x0.data(), u0.data());
    nmpc_amd::DDPSolverSharded whole(model, T, B, {0});
    whole.config().max_iter = 5;
    whole.solve(t0.data(), x0.data(), u0.data());
    int bad = 0;
    long long it = 0;
    for(int b = 0; b < B; b++)
    {
      bad += std::memcmp(sharded.record(b), whole.record(b), sharded.recordWidth() * sizeof(double)) != 0;
      bad += sharded.status(b) != whole.status(b) || sharded.iters(b) != whole.iters(b);
      it += sharded.iters(b);
    }
    std::printf("%s: %d instances over %d shards (%s gather): %lld iterations, %d instances differ from the unsharded solve\n", model.c_str(), B,
                S, rccl ? "RCCL all-gather" : "peer-copy", it, bad);
    if(bad == 0 && it > 0)
    {
      std::printf("SHARDED_OK\n");
    }
    return bad != 0;
  }
  catch(const std::exception & e)
  {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
}
