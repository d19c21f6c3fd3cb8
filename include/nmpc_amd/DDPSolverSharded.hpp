// One node, several MI355X, one host process: a batch of independent DDP problems sharded over the GPUs.
//
// Every instance is a self-contained solve (the reference keeps all solver state per object,
// nmpc_ddp/include/nmpc_ddp/DDPSolver.h:329-374), so the batch splits into contiguous shards — the same split as
// nmpc_amd/sharding.py: shard s of S owns [s B / S + min(s, B % S), ...) — one solver handle per shard, each on its own
// device and stream; the shards' persistent kernels run concurrently and nothing is exchanged during the iterations.  The
// ONE collective of a job is the final gather of the packed result records [X | U | cost] (status and iteration count, two
// ints per instance, go to the host directly):
//   Gather::Rccl   ncclAllGather over xGMI (one communicator per device, ncclCommInitAll), shards padded to the largest;
//                  every device ends up with all records, device 0's copy goes to the host.  Needs distinct devices.
//   Gather::Copy   hipMemcpyPeerAsync of every shard's records into the root device's buffer (point-to-point over the same
//                  links; also works when several shards share a device, which is how the one-GPU test box runs it).
// The torch.distributed path of bench.py (one process per GPU, RCCL all_gather) is the multi-process twin of this helper.
//
// Works on the C-ABI (include/nmpc_hip_ddp.h) with flat arrays in the reference layouts; link with -lnmpc_hip_ddp
// -lamdhip64 and, for Gather::Rccl, define NMPC_AMD_WITH_RCCL and add -lrccl.
#pragma once

#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <nmpc_hip_ddp.h>
#ifdef NMPC_AMD_WITH_RCCL
#  include <rccl/rccl.h>
#endif

namespace nmpc_amd
{
/** Contiguous, balanced slice [lo, hi) of `batch` instances owned by shard `rank` of `world` (sizes differ by at most 1). */
inline void shardRange(int batch, int rank, int world, int & lo, int & hi)
{
  const int base = batch / world, rem = batch % world;
  lo = rank * base + (rank < rem ? rank : rem);
  hi = lo + base + (rank < rem ? 1 : 0);
}

class DDPSolverSharded
{
public:
  enum class Gather
  {
    Copy,
    Rccl
  };

  /** \param model registry name of the problem type (nmpc_hip_ddp_model_name)
      \param devices HIP device of every shard (Gather::Rccl: pairwise distinct) */
  DDPSolverSharded(const std::string & model, int horizon_steps, int batch, const std::vector<int> & devices, Gather gather = Gather::Copy)
  : batch_(batch), T_(horizon_steps), gather_(gather)
  {
    if(devices.empty() || batch < static_cast<int>(devices.size()))
    {
      throw std::invalid_argument("need at least one device and one instance per shard");
    }
    try
    {
      construct(model, batch, devices);
    }
    catch(...)
    {
      release(); // the destructor of a half-built object never runs: give back what was created so far
      throw;
    }
  }

private:
  void construct(const std::string & model, int batch, const std::vector<int> & devices)
  {
    size_t param_bytes = 0;
    int dyn = 0;
    check(nmpc_hip_ddp_model_info(model.c_str(), &n_, &m_, &dyn, &param_bytes));
    mm_ = m_ > 0 ? m_ : 1;
    width_ = static_cast<size_t>(T_ + 1) * n_ + static_cast<size_t>(T_) * mm_ + (T_ + 1);
    status_.assign(batch, 0);
    iters_.assign(batch, 0);
    shards_.resize(devices.size());
    for(size_t s = 0; s < shards_.size(); s++)
    {
      Shard & sh = shards_[s];
      sh.device = devices[s];
      shardRange(batch, static_cast<int>(s), static_cast<int>(shards_.size()), sh.lo, sh.hi);
      max_shard_ = std::max(max_shard_, sh.hi - sh.lo);
      check(nmpc_hip_ddp_create(model.c_str(), T_, sh.hi - sh.lo, sh.device, &sh.handle));
      // every shard runs on the kernel family the WHOLE batch would get: shards == the unsharded solve, bit for bit (in fp32 the
      // family depends on the batch size; families agree in their decisions, not in the last bits of their values)
      check(nmpc_hip_ddp_set_dispatch_batch(sh.handle, batch));
    }
    for(Shard & sh : shards_)
    {
      const size_t nb = static_cast<size_t>(sh.hi - sh.lo);
      hipCheck(hipSetDevice(sh.device));
      hipCheck(hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking));
      hipCheck(hipMalloc(reinterpret_cast<void **>(&sh.d_in), (nb * (1 + n_ + static_cast<size_t>(T_) * mm_)) * sizeof(double)));
      // send buffer padded to the largest shard (a fixed-size all-gather); the gathered records on every device (Rccl) or on
      // the root only (Copy)
      hipCheck(hipMalloc(reinterpret_cast<void **>(&sh.d_rec), static_cast<size_t>(max_shard_) * width_ * sizeof(double)));
      hipCheck(hipMemset(sh.d_rec, 0, static_cast<size_t>(max_shard_) * width_ * sizeof(double)));
      if(gather_ == Gather::Rccl || &sh == &shards_[0])
      {
        hipCheck(hipMalloc(reinterpret_cast<void **>(&sh.d_all), shards_.size() * max_shard_ * width_ * sizeof(double)));
      }
    }
    nmpc_hip_ddp_default_config(&config_);
    config_.horizon_steps = T_;
    if(gather_ == Gather::Rccl)
    {
#ifdef NMPC_AMD_WITH_RCCL
      comms_.resize(shards_.size());
      if(ncclCommInitAll(comms_.data(), static_cast<int>(devices.size()), devices.data()) != ncclSuccess)
      {
        throw std::runtime_error("ncclCommInitAll failed (Gather::Rccl needs one distinct device per shard)");
      }
#else
      throw std::runtime_error("built without NMPC_AMD_WITH_RCCL: use Gather::Copy");
#endif
    }
  }

  /** Gives back every handle, stream, buffer and communicator (idempotent). */
  void release()
  {
#ifdef NMPC_AMD_WITH_RCCL
    for(ncclComm_t c : comms_)
    {
      if(c)
      {
        ncclCommDestroy(c);
      }
    }
    comms_.clear();
#endif
    for(Shard & sh : shards_)
    {
      (void)hipSetDevice(sh.device);
      if(sh.stream)
      {
        (void)hipStreamSynchronize(sh.stream);
        (void)hipStreamDestroy(sh.stream);
      }
      (void)hipFree(sh.d_in);
      (void)hipFree(sh.d_rec);
      (void)hipFree(sh.d_all);
      (void)hipFree(sh.scratch);
      if(sh.handle)
      {
        nmpc_hip_ddp_destroy(sh.handle);
      }
    }
    shards_.clear();
  }

public:
  ~DDPSolverSharded()
  {
    release();
  }
  DDPSolverSharded(const DDPSolverSharded &) = delete;
  DDPSolverSharded & operator=(const DDPSolverSharded &) = delete;

  /** DDPSolver::config() of every shard. */
  nmpc_hip_ddp_config & config()
  {
    return config_;
  }
  int shards() const
  {
    return static_cast<int>(shards_.size());
  }
  size_t recordWidth() const
  {
    return width_;
  }

  /** DDPSolver::solve for the whole batch: t0[B] (or nullptr), x0[B][n], u_init[B][T][MM] on the host.  The shards' kernels
      are queued without waiting for one another; then the one gather.  Results: X(), U(), cost(), status(), iters(). */
  void solve(const double * t0, const double * x0, const double * u_init)
  {
    const size_t nu = static_cast<size_t>(T_) * mm_;
    for(Shard & sh : shards_)
    {
      const size_t nb = static_cast<size_t>(sh.hi - sh.lo);
      hipCheck(hipSetDevice(sh.device));
      check(nmpc_hip_ddp_set_config(sh.handle, &config_));
      double * d_t0 = sh.d_in;
      double * d_x0 = d_t0 + nb;
      double * d_u0 = d_x0 + nb * n_;
      if(t0)
      {
        hipCheck(hipMemcpyAsync(d_t0, t0 + sh.lo, nb * sizeof(double), hipMemcpyHostToDevice, sh.stream));
      }
      hipCheck(hipMemcpyAsync(d_x0, x0 + static_cast<size_t>(sh.lo) * n_, nb * n_ * sizeof(double), hipMemcpyHostToDevice, sh.stream));
      hipCheck(hipMemcpyAsync(d_u0, u_init + static_cast<size_t>(sh.lo) * nu, nb * nu * sizeof(double), hipMemcpyHostToDevice, sh.stream));
      check(nmpc_hip_ddp_solve_device(sh.handle, t0 ? d_t0 : nullptr, d_x0, d_u0, sh.stream));
      // pack [X | U | cost] of the shard on its own stream behind the solve: no host synchronisation between the shards
      packRecords(sh);
    }
    gatherRecords();
  }

  //! gathered record of instance b: [X (T+1) n | U T MM | cost T+1]
  const double * record(int b) const
  {
    const Shard & sh = shards_[shardOf(b)];
    const size_t s = static_cast<size_t>(&sh - shards_.data());
    return all_.data() + (s * max_shard_ + static_cast<size_t>(b - sh.lo)) * width_;
  }
  const double * X(int b) const
  {
    return record(b);
  }
  const double * U(int b) const
  {
    return record(b) + static_cast<size_t>(T_ + 1) * n_;
  }
  const double * cost(int b) const
  {
    return U(b) + static_cast<size_t>(T_) * mm_;
  }
  int status(int b) const
  {
    return status_[b];
  }
  int iters(int b) const
  {
    return iters_[b];
  }

private:
  struct Shard
  {
    int device = 0, lo = 0, hi = 0;
    nmpc_hip_ddp_handle handle = nullptr;
    hipStream_t stream = nullptr;
    double * d_in = nullptr; //!< t0 | x0 | u_init of the shard
    double * d_rec = nullptr; //!< packed records of the shard, padded to the largest shard
    double * d_all = nullptr; //!< gathered records (every device with Rccl, the root with Copy)
    void * scratch = nullptr; //!< the C-ABI's per-field blocks before they are interleaved into records
    size_t scratch_bytes = 0;
  };

  int shardOf(int b) const
  {
    for(size_t s = 0; s < shards_.size(); s++)
    {
      if(b >= shards_[s].lo && b < shards_[s].hi)
      {
        return static_cast<int>(s);
      }
    }
    throw std::out_of_range("instance index");
  }

  /** The shard's result fields, instance-major, into d_rec — field by field with strided device copies (the C-ABI returns
      every field as one [B][...] block). */
  void packRecords(Shard & sh)
  {
    const size_t nb = static_cast<size_t>(sh.hi - sh.lo);
    const size_t nx = static_cast<size_t>(T_ + 1) * n_, nu = static_cast<size_t>(T_) * mm_, nc = static_cast<size_t>(T_ + 1);
    ensureScratch(sh, nb * (nx + nu + nc) * sizeof(double) + 2 * nb * sizeof(int));
    double * fx = static_cast<double *>(sh.scratch);
    double * fu = fx + nb * nx;
    double * fc = fu + nb * nu;
    int * fs = reinterpret_cast<int *>(fc + nb * nc);
    int * fi = fs + nb;
    check(nmpc_hip_ddp_get_device(sh.handle, NMPC_HIP_FIELD_X, fx, nb * nx * sizeof(double), sh.stream));
    check(nmpc_hip_ddp_get_device(sh.handle, NMPC_HIP_FIELD_U, fu, nb * nu * sizeof(double), sh.stream));
    check(nmpc_hip_ddp_get_device(sh.handle, NMPC_HIP_FIELD_COST, fc, nb * nc * sizeof(double), sh.stream));
    check(nmpc_hip_ddp_get_device(sh.handle, NMPC_HIP_FIELD_STATUS, fs, nb * sizeof(int), sh.stream));
    check(nmpc_hip_ddp_get_device(sh.handle, NMPC_HIP_FIELD_ITERS, fi, nb * sizeof(int), sh.stream));
    const size_t wb = width_ * sizeof(double);
    hipCheck(hipMemcpy2DAsync(sh.d_rec, wb, fx, nx * sizeof(double), nx * sizeof(double), nb, hipMemcpyDeviceToDevice, sh.stream));
    hipCheck(hipMemcpy2DAsync(sh.d_rec + nx, wb, fu, nu * sizeof(double), nu * sizeof(double), nb, hipMemcpyDeviceToDevice, sh.stream));
    hipCheck(hipMemcpy2DAsync(sh.d_rec + nx + nu, wb, fc, nc * sizeof(double), nc * sizeof(double), nb, hipMemcpyDeviceToDevice, sh.stream));
    hipCheck(hipMemcpyAsync(status_.data() + sh.lo, fs, nb * sizeof(int), hipMemcpyDeviceToHost, sh.stream));
    hipCheck(hipMemcpyAsync(iters_.data() + sh.lo, fi, nb * sizeof(int), hipMemcpyDeviceToHost, sh.stream));
  }

  void gatherRecords()
  {
    const size_t shard_bytes = static_cast<size_t>(max_shard_) * width_ * sizeof(double);
    all_.resize(shards_.size() * max_shard_ * width_);
    if(gather_ == Gather::Rccl)
    {
#ifdef NMPC_AMD_WITH_RCCL
      ncclGroupStart();
      for(size_t s = 0; s < shards_.size(); s++)
      {
        hipCheck(hipSetDevice(shards_[s].device));
        ncclAllGather(shards_[s].d_rec, shards_[s].d_all, static_cast<size_t>(max_shard_) * width_, ncclDouble, comms_[s], shards_[s].stream);
      }
      ncclGroupEnd();
      for(Shard & sh : shards_)
      {
        hipCheck(hipSetDevice(sh.device));
        hipCheck(hipStreamSynchronize(sh.stream));
      }
#endif
    }
    else
    {
      Shard & root = shards_[0];
      for(size_t s = 0; s < shards_.size(); s++)
      {
        Shard & sh = shards_[s];
        hipCheck(hipSetDevice(sh.device));
        hipCheck(hipMemcpyPeerAsync(reinterpret_cast<char *>(root.d_all) + s * shard_bytes, root.device, sh.d_rec, sh.device, shard_bytes,
                                    sh.stream));
      }
      for(Shard & sh : shards_)
      {
        hipCheck(hipSetDevice(sh.device));
        hipCheck(hipStreamSynchronize(sh.stream));
      }
    }
    hipCheck(hipSetDevice(shards_[0].device));
    hipCheck(hipMemcpy(all_.data(), shards_[0].d_all, all_.size() * sizeof(double), hipMemcpyDeviceToHost));
  }

  void ensureScratch(Shard & sh, size_t bytes)
  {
    if(sh.scratch_bytes < bytes)
    {
      if(sh.scratch)
      {
        hipCheck(hipFree(sh.scratch));
      }
      hipCheck(hipMalloc(&sh.scratch, bytes));
      sh.scratch_bytes = bytes;
    }
  }

  static void check(int rc)
  {
    if(rc == NMPC_HIP_ERR_INVALID_ARGUMENT)
    {
      throw std::invalid_argument(nmpc_hip_ddp_last_error());
    }
    if(rc != NMPC_HIP_OK)
    {
      throw std::runtime_error(nmpc_hip_ddp_last_error());
    }
  }
  static void hipCheck(hipError_t e)
  {
    if(e != hipSuccess)
    {
      throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e));
    }
  }

  int batch_, T_, n_ = 0, m_ = 0, mm_ = 1, max_shard_ = 0;
  size_t width_ = 0;
  Gather gather_;
  nmpc_hip_ddp_config config_;
  std::vector<Shard> shards_;
  std::vector<double> all_;
  std::vector<int> status_, iters_;
#ifdef NMPC_AMD_WITH_RCCL
  std::vector<ncclComm_t> comms_;
#endif
};
} // namespace nmpc_amd
