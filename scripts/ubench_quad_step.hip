// "Step lab" for the backward recursion of the quad kernel (include/nmpc_amd/hip/ddp_kernels_quad.hpp, backwardQuad): one
// wavefront runs the Riccati step of four cart-pole-shaped instances (n = 4, m = 1) on v_mfma_f64_4x4x4 from derivative
// records in LDS, exactly the instruction mix of the kernel's straight-line chunk, and reports shader cycles per timestep
// for several ORDERINGS of the same arithmetic.  Used to decide how to schedule the step (the machine issues in order).
//   hipcc --offload-arch=gfx950 -O3 -mllvm --amdgpu-mfma-vgpr-form scripts/ubench_quad_step.hip -o scripts/ubench_quad_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int oFx = 0, oLxx = 16, oFu = 32, oLxu = 36, oLx = 40, oLuu = 44, oLu = 45, oU = 46, oZero = 47, oUinv = 48, kRecQ = 49;
constexpr int kChunkSteps = 16, kGainRec = 7;
constexpr int kDummy = 192; // doubles per wave: every lane's own slot for a store it does not mean (+ 16 timesteps x 7)
constexpr int kWaveDoubles = 64 * (kRecQ + kGainRec) + kDummy;
constexpr int gK = 0, gKfb = 1, gLive = 5, gDummy = 6;

__device__ inline double mma(double a, double b, double c)
{
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
template<int kSrc>
__device__ inline double quadBroadcast(double v)
{
  constexpr int ctrl = kSrc | (kSrc << 2) | (kSrc << 4) | (kSrc << 6);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double recipFast(double x)
{
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ inline double pick(bool p, double v)
{
  return p ? v : 0.0;
}

struct Operands
{
  double Fx, Lxx, LxxT, FuM, FuB, LM, CM, LxuRow, uinv;
};

/** VARIANT 0: the kernel's order today.  1: Vx path split off the Vxx recurrence (two more MFMAs) and deferred by one
    timestep, its instructions placed between the next step's MFMAs.  2: as 1 but the deferred block placed after the
    MFMAs (control: does placement matter?).  3: as 0 without the off-chain work at all (lower bound of the Vxx chain). */
template<int VARIANT, int FEAT = 15>
__global__ __launch_bounds__(256) void step_lab(double * out, long long * cyc, int n_chunks, double lam, double * gout)
{
  extern __shared__ double lds[];
  const int wave = threadIdx.x / 64, wl = threadIdx.x % 64;
  double * chunk = lds + wave * kWaveDoubles;
  const int row = wl / 16, blk = (wl / 4) % 4, col = wl % 4;
  // synthetic records: a stable Riccati recursion (Fx ~ 0.95 I + small, Lxx = I, Luu = 1)
  {
    double * rec = chunk + wl * kRecQ;
    for(int r = 0; r < 4; r++)
    {
      for(int c = 0; c < 4; c++)
      {
        rec[oFx + 4 * r + c] = (r == c ? 0.95 : 0.01 * (r + 1)) + 1e-3 * (wl % 7);
        rec[oLxx + 4 * r + c] = (r == c ? 1.0 : 0.0);
      }
      rec[oFu + r] = 0.1 * (r + 1);
      rec[oLxu + r] = 0.0;
      rec[oLx + r] = 0.01 * r;
    }
    rec[oLuu] = 1.0;
    rec[oLu] = 0.02;
    rec[oU] = 0.3;
    rec[oZero] = 0.0;
    rec[oUinv] = 1.0 / 1.3;
  }
  __syncthreads();
  double * gains = chunk + 64 * kRecQ;
  double * gain_q = gains + (blk * 16) * kGainRec
                    + (row == 0 ? gKfb + col : ((row == 1 && col == 0) ? gK : ((row == 1 && col == 1) ? gLive : gDummy)));
  const double * rec_q = chunk + (blk * 16) * kRecQ;
  const bool c0 = col == 0, c1 = col == 1, r0 = row == 0;
  const int aE = 4 * row + col, aT = 4 * col + row;
  const int aFuM = c0 ? oFu + row : oZero, aFuB = oFu + row;
  const int aLM = c0 ? oLxu + row : (c1 ? oLx + row : oZero);
  const int aLMx = c0 ? oLxu + row : oZero; // [Lxu | 0 | 0 | 0]
  const int aLMv = c1 ? oLx + row : oZero; // [0 | Lx | 0 | 0]
  const int aCM = c0 ? oLuu : (c1 ? oLu : oZero);
  const int aLxuRow = oLxu + col;
  (void)aLMx;
  (void)aLMv;
  auto loadOperands = [&](int ts, Operands & o)
  {
    const double * R = rec_q + ts * kRecQ;
    o.Fx = R[oFx + aE];
    o.Lxx = R[oLxx + aE];
    o.LxxT = R[oLxx + aT];
    o.FuM = R[aFuM];
    o.FuB = R[aFuB];
    o.LM = R[aLM];
    o.CM = R[aCM];
    o.LxuRow = R[aLxuRow];
    o.uinv = R[oUinv];
  };
  double Vxx = (row == col) ? 1.0 : 0.0;
  double VxM = pick(c1, 0.1 * row);
  double dV0_l = 0, dV1_l = 0, krn = 0;
  bool ok = true;
  const bool need = lam >= 0; // runtime-true

  // ---- variant 0 / 3
  double * gain_u = gains + (blk * 16) * kGainRec + (row == 0 ? gKfb + col : ((row == 1 && col == 0) ? gK : gDummy));
  auto step0 = [&](int ts, const Operands & o, int ts_next, Operands & o_next)
  {
    if(FEAT & 1)
    {
      loadOperands(ts_next, o_next);
    }
    else
    {
      o_next = o;
    }
    const double P = mma(Vxx, o.Fx, 0.0);
    const double Rm = mma(Vxx, o.FuM, VxM);
    const double Qxx = mma(P, o.Fx, o.Lxx);
    const double QxxT = mma(o.Fx, P, o.LxxT);
    const double S = mma(o.Fx, Rm, o.LM);
    const double Wq = mma(o.FuB, Rm, o.CM);
    const double QA = mma(quadBroadcast<0>(Rm), o.Fx, o.LxuRow);
    const double Quu = quadBroadcast<0>(Wq);
    const double Qu = quadBroadcast<1>(Wq);
    const double Qr = quadBroadcast<0>(S);
    const double Qxr = quadBroadcast<1>(S);
    const double Quu_F = Quu + lam;
    const bool step_ok = !(Quu_F <= 0);
    const double inv = (VARIANT == 16 || step_ok) ? recipFast(Quu_F) : 0.0;
    const double k = -1 * (Qu * inv);
    const double Kc = -1 * (QA * inv);
    const double Kr = -1 * (Qr * inv);
    const bool live = need && ok && step_ok;
    ok = ok && step_ok;
    if(VARIANT != 3 && (FEAT & 2))
    {
      if(VARIANT == 16 || live)
      {
        dV0_l += k * Qu;
        dV1_l += 0.5 * (k * (Quu * k));
      }
    }
    const double KQr = Kr * Quu, KQc = Kc * Quu;
    const double Vn = fma(Qr, Kc, fma(Kr, QA, fma(KQr, Kc, Qxx)));
    const double VnT = fma(QA, Kr, fma(Kc, Qr, fma(KQc, Kr, QxxT)));
    Vxx = 0.5 * (Vn + VnT);
    VxM = pick(c1, fma(Qr, k, fma(Kr, Qu, fma(KQr, k, Qxr))));
    if(VARIANT == 16)
    {
      gain_u[ts * kGainRec] = r0 ? Kc : k;
    }
    else if(VARIANT != 3 && (FEAT & 4))
    {
      gain_q[ts * kGainRec] = r0 ? Kc : (c0 ? k : (live ? 1.0 : 0.0));
    }
    if(VARIANT != 3 && VARIANT != 16 && (FEAT & 8))
    {
      krn = fmax(krn, fabs(k) * o.uinv);
    }
  };

  // ---- variants 1 / 2: the Vx / k / dV / gains part of a step runs one step late
  struct Deferred
  {
    double Fx, FuB, LM, CM, uinv; // operands of that step
    double inv, Kc, Kr, KQr, Qr, Quu;
    bool live;
    int ts;
  };
  Deferred df;
  bool have_df = false;
  double t_Sb = 0, t_Wqb = 0;
  auto tailMfma = [&](const Deferred & d)
  {
    t_Sb = mma(d.Fx, VxM, d.LM); // column 1: Qx
    t_Wqb = mma(d.FuB, VxM, d.CM); // column 1: Qu
  };
  auto tailValu = [&](const Deferred & d)
  {
    const double Qu = quadBroadcast<1>(t_Wqb);
    const double Qxr = quadBroadcast<1>(t_Sb);
    const double k = -1 * (Qu * d.inv);
    if(d.live)
    {
      dV0_l += k * Qu;
      dV1_l += 0.5 * (k * (d.Quu * k));
    }
    VxM = pick(c1, fma(d.Qr, k, fma(d.Kr, Qu, fma(d.KQr, k, Qxr))));
    gain_q[d.ts * kGainRec] = r0 ? d.Kc : (c0 ? k : (d.live ? 1.0 : 0.0));
    krn = fmax(krn, fabs(k) * d.uinv);
  };
  auto step1 = [&](int ts, const Operands & o, int ts_next, Operands & o_next)
  {
    loadOperands(ts_next, o_next);
    const double P = mma(Vxx, o.Fx, 0.0);
    const double Rx = mma(Vxx, o.FuM, 0.0); // [(Fu^T Vxx)^T | 0 | 0 | 0]
    if(VARIANT == 1 && have_df)
    {
      tailMfma(df);
    }
    const double Qxx = mma(P, o.Fx, o.Lxx);
    const double QxxT = mma(o.Fx, P, o.LxxT);
    const double S = mma(o.Fx, Rx, o.LM); // column 0 = Qux^T
    const double Wq = mma(o.FuB, Rx, o.CM); // column 0 = Quu
    const double QA = mma(quadBroadcast<0>(Rx), o.Fx, o.LxuRow);
    if(VARIANT == 1 && have_df)
    {
      tailValu(df);
    }
    const double Quu = quadBroadcast<0>(Wq);
    const double Qr = quadBroadcast<0>(S);
    const double Quu_F = Quu + lam;
    const bool step_ok = !(Quu_F <= 0);
    const double inv = step_ok ? recipFast(Quu_F) : 0.0;
    const double Kc = -1 * (QA * inv);
    const double Kr = -1 * (Qr * inv);
    const bool live = need && ok && step_ok;
    ok = ok && step_ok;
    const double KQr = Kr * Quu, KQc = Kc * Quu;
    const double Vn = fma(Qr, Kc, fma(Kr, QA, fma(KQr, Kc, Qxx)));
    const double VnT = fma(QA, Kr, fma(Kc, Qr, fma(KQc, Kr, QxxT)));
    Vxx = 0.5 * (Vn + VnT);
    if(VARIANT == 2 && have_df)
    {
      tailMfma(df);
      tailValu(df);
    }
    df.Fx = o.Fx;
    df.FuB = o.FuB;
    df.LM = o.LM;
    df.CM = o.CM;
    df.uinv = o.uinv;
    df.inv = inv;
    df.Kc = Kc;
    df.Kr = Kr;
    df.KQr = KQr;
    df.Qr = Qr;
    df.Quu = Quu;
    df.live = live;
    df.ts = ts;
    have_df = true;
  };


  // ---- variants 4 / 5 / 6: no lane broadcasts at all.  Fu, Lxu, Lx enter as "row-broadcast" operands (entry [row] in every
  // column), Luu, Lu as constants, so that the matrix core leaves Quu, Qu, Qux[row], Qux[col], Qx[row] in EVERY lane that
  // needs them; the Vx path has its own two MFMAs (same products in the same order: same bits as variant 0).
  // 4: Vx path right after the Vxx chain; 5: deferred by one step into the next MFMA phase; 6: as 4 without gating / krn
  struct OperandsB
  {
    double Fx, Lxx, LxxT, FuB, LxuB, LxB, LxuRow, Luu, Lu, uinv;
  };
  const int aRow = row;
  int qFx = blk * 16 * kRecQ + oFx + aE, qLxx = blk * 16 * kRecQ + oLxx + aE, qLxxT = blk * 16 * kRecQ + oLxx + aT;
  int qFu = blk * 16 * kRecQ + oFu + row, qLxu = blk * 16 * kRecQ + oLxu + row, qLx = blk * 16 * kRecQ + oLx + row;
  int qLxuRow = blk * 16 * kRecQ + oLxu + col, qLuu = blk * 16 * kRecQ + oLuu, qLu = blk * 16 * kRecQ + oLu, qUinv = blk * 16 * kRecQ + oUinv;
  asm volatile("" : "+v"(qFx), "+v"(qLxx), "+v"(qLxxT), "+v"(qFu), "+v"(qLxu), "+v"(qLx), "+v"(qLxuRow), "+v"(qLuu), "+v"(qLu), "+v"(qUinv));
  auto loadOperandsB = [&](int ts, OperandsB & o)
  {
    if constexpr(VARIANT >= 8 && VARIANT != 12 && VARIANT != 13 && VARIANT != 14 && VARIANT != 15)
    {
      // one ds_read_b64 each (a ds_read2_b64 costs three): every operand has its own opaque base offset
      const double * R = chunk + ts * kRecQ;
      o.Fx = R[qFx];
      o.Lxx = R[qLxx];
      o.LxxT = R[qLxxT];
      o.FuB = R[qFu];
      o.LxuB = R[qLxu];
      o.LxB = R[qLx];
      o.LxuRow = R[qLxuRow];
      o.Luu = R[qLuu];
      o.Lu = R[qLu];
      o.uinv = R[qUinv];
      return;
    }
    const double * R = rec_q + ts * kRecQ;
    o.Fx = R[oFx + aE];
    o.Lxx = R[oLxx + aE];
    o.LxxT = R[oLxx + aT];
    o.FuB = R[oFu + aRow];
    o.LxuB = R[oLxu + aRow];
    o.LxB = R[oLx + aRow];
    o.LxuRow = R[oLxu + col];
    o.Luu = R[oLuu];
    o.Lu = R[oLu];
    if(VARIANT < 6 || VARIANT == 12 || VARIANT == 13 || VARIANT == 14)
    {
      o.uinv = R[oUinv];
    }
  };
  double VxB = 0.1 * row; // Vx[row] in every column
  double * dummy = chunk + 64 * (kRecQ + kGainRec) + wl;
  double * gain_K = (VARIANT >= 7) ? (row == 0 ? gains + (blk * 16) * kGainRec + gKfb + col : dummy)
                                                   : gains + (blk * 16) * kGainRec + (row == 0 ? gKfb + col : gDummy);
  double * gain_k = (VARIANT >= 7) ? ((row == 1 && col == 0) ? gains + (blk * 16) * kGainRec + gK : dummy)
                                                   : gains + (blk * 16) * kGainRec + ((row == 1 && col == 0) ? gK : gDummy);
  int n_live = 0;
  struct DeferredB
  {
    double Fx, FuB, LxB, Lu, uinv, ninv, Kr, KQr, Qr, Quu, Kc;
    int ts;
  };
  DeferredB dB;
  double hK[kChunkSteps], hk[kChunkSteps];
  auto vxPath = [&](const DeferredB & d)
  {
    const double Qxr = mma(d.Fx, VxB, d.LxB); // Qx[row] in every column
    const double Qu = mma(d.FuB, VxB, d.Lu); // Qu in every lane
    const double k = Qu * d.ninv;
    dV0_l += k * Qu;
    dV1_l += 0.5 * (k * (d.Quu * k));
    VxB = fma(d.Qr, k, fma(d.Kr, Qu, fma(d.KQr, k, Qxr)));
    if constexpr(VARIANT == 15)
    {
      gain_u[d.ts * kGainRec] = r0 ? d.Kc : k;
      return;
    }
    if constexpr(VARIANT >= 8)
    {
      hk[d.ts] = k;
      krn = fmax(krn, fabs(k) * d.uinv);
      return;
    }
    if(FEAT & 4)
    {
      gain_k[d.ts * kGainRec] = k;
    }
    if(VARIANT < 6)
    {
      krn = fmax(krn, fabs(k) * d.uinv);
    }
  };
  auto stepBody = [&](int ts, const OperandsB & o)
  {
    const double P = mma(Vxx, o.Fx, 0.0);
    const double Rx = mma(Vxx, o.FuB, 0.0); // (Vxx Fu)[row] in every column
    const double Qxx = mma(P, o.Fx, o.Lxx);
    const double QxxT = mma(o.Fx, P, o.LxxT);
    if((VARIANT == 5 || VARIANT == 10) && have_df)
    {
      vxPath(dB);
    }
    const double Quu = mma(o.FuB, Rx, o.Luu); // everywhere
    const double Qr = mma(o.Fx, Rx, o.LxuB); // Qux[row] in every column
    const double QA = mma(Rx, o.Fx, o.LxuRow); // Qux[col] in every row
    const double Quu_F = Quu + lam;
    const bool step_ok = !(Quu_F <= 0);
    const double ninv = -recipFast(Quu_F);
    const double Kc = QA * ninv;
    const double Kr = Qr * ninv;
    ok = ok && step_ok;
    if constexpr(VARIANT < 8)
    {
      n_live += ok ? 1 : 0;
    }
    const double KQr = Kr * Quu, KQc = Kc * Quu;
    const double Vn = fma(Qr, Kc, fma(Kr, QA, fma(KQr, Kc, Qxx)));
    const double VnT = fma(QA, Kr, fma(Kc, Qr, fma(KQc, Kr, QxxT)));
    Vxx = 0.5 * (Vn + VnT);
    if constexpr(VARIANT == 15)
    {
    }
    else if constexpr(VARIANT >= 8)
    {
      hK[ts] = Kc;
    }
    else if(FEAT & 4)
    {
      gain_K[ts * kGainRec] = Kc;
    }
    dB.Fx = o.Fx;
    dB.FuB = o.FuB;
    dB.LxB = o.LxB;
    dB.Lu = o.Lu;
    dB.uinv = o.uinv;
    dB.ninv = ninv;
    dB.Kr = Kr;
    dB.KQr = KQr;
    dB.Qr = Qr;
    dB.Quu = Quu;
    dB.Kc = Kc;
    dB.ts = ts;
    have_df = true;
    if(VARIANT != 5 && VARIANT != 10)
    {
      vxPath(dB);
    }
  };
  auto stepB = [&](int ts, const OperandsB & o, int ts_next, OperandsB & o_next)
  {
    if(FEAT & 1)
    {
      loadOperandsB(ts_next, o_next);
    }
    else
    {
      o_next = o;
    }
    stepBody(ts, o);
  };

  auto stepB14 = [&](int ts, const OperandsB & o)
  {
    OperandsB unused;
    stepBody(ts, o);
    (void)unused;
  };
  const long long t0 = __builtin_readcyclecounter();
  if constexpr(VARIANT >= 4)
  {
    for(int ch = 0; ch < n_chunks; ch++)
    {
      if constexpr(VARIANT == 14)
      {
        // the kernel's request pattern: operands of two timesteps together, two timesteps ahead
        OperandsB o[kChunkSteps], dummy_o;
        loadOperandsB(kChunkSteps - 1, o[kChunkSteps - 1]);
        loadOperandsB(kChunkSteps - 2, o[kChunkSteps - 2]);
#pragma unroll
        for(int ts = kChunkSteps - 1; ts >= 0; ts--)
        {
          if((ts & 1) && ts >= 3)
          {
            loadOperandsB(ts - 2, o[ts - 2]);
            loadOperandsB(ts - 3, o[ts - 3]);
          }
          stepB14(ts, o[ts]);
        }
        (void)dummy_o;
      }
      else
      {
      OperandsB o2[2];
      loadOperandsB(kChunkSteps - 1, o2[(kChunkSteps - 1) & 1]);
#pragma unroll
      for(int ts = kChunkSteps - 1; ts >= 0; ts--)
      {
        stepB(ts, o2[ts & 1], ts > 0 ? ts - 1 : 0, o2[(ts & 1) ^ 1]);
      }
      }
      if constexpr(VARIANT == 11)
      {
#pragma unroll
        for(int ts = 0; ts < kChunkSteps; ts++)
        {
          krn += hK[ts] + hk[ts];
        }
      }
      else if constexpr(VARIANT == 8 || VARIANT == 10 || VARIANT == 12 || VARIANT == 14)
      {
        // gains straight to HBM from the lanes that hold them, once per chunk
        double * gK_out = gout + (blockIdx.x * 4 + wave) * 8192 + (ch & 3) * 2048;
        if(row == 0)
        {
#pragma unroll
          for(int ts = 0; ts < kChunkSteps; ts++)
          {
            gK_out[(ts * 4 + col) * 16 + blk] = hK[ts];
          }
        }
        if(row == 1 && col == 0)
        {
#pragma unroll
          for(int ts = 0; ts < kChunkSteps; ts++)
          {
            gK_out[1024 + ts * 16 + blk] = hk[ts];
          }
        }
      }
      else if constexpr(VARIANT == 15)
      {
      }
      else if constexpr(VARIANT == 9 || VARIANT == 13)
      {
#pragma unroll
        for(int ts = 0; ts < kChunkSteps; ts++)
        {
          gain_K[ts * kGainRec] = hK[ts];
          gain_k[ts * kGainRec] = hk[ts];
        }
      }
      else
      {
        gain_k[kGainRec * 15 + 3] = n_live; // (the live count goes out once per chunk)
      }
    }
    if(VARIANT == 5 || VARIANT == 10)
    {
      vxPath(dB);
    }
    VxM = VxB;
  }
  else
  for(int ch = 0; ch < n_chunks; ch++)
  {
    Operands o2[2];
    loadOperands(kChunkSteps - 1, o2[(kChunkSteps - 1) & 1]);
#pragma unroll
    for(int ts = kChunkSteps - 1; ts >= 0; ts--)
    {
      if constexpr(VARIANT == 0 || VARIANT == 3 || VARIANT == 16)
      {
        step0(ts, o2[ts & 1], ts > 0 ? ts - 1 : 0, o2[(ts & 1) ^ 1]);
      }
      else
      {
        step1(ts, o2[ts & 1], ts > 0 ? ts - 1 : 0, o2[(ts & 1) ^ 1]);
      }
    }
  }
  if constexpr(VARIANT == 1 || VARIANT == 2)
  {
    tailMfma(df);
    tailValu(df);
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = Vxx + VxM + dV0_l + dV1_l + krn + (ok ? 1 : 0) + gains[wl];
  if(wl == 0)
  {
    cyc[blockIdx.x * 4 + wave] = t1 - t0;
  }
}

template<int V, int FEAT = 15>
void run(const char * name, int waves, int blocks = 1)
{
  double * out;
  double * gout;
  long long * cyc;
  (void)hipMalloc(&gout, static_cast<size_t>(blocks) * 4 * 8192 * 8);
  (void)hipMalloc(&out, static_cast<size_t>(blocks) * 256 * 8);
  (void)hipMalloc(&cyc, static_cast<size_t>(blocks) * 4 * 8);
  const int n_chunks = 2000;
  const size_t lds = 4 * kWaveDoubles * sizeof(double);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(step_lab<V, FEAT>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  step_lab<V, FEAT><<<blocks, waves * 64, lds>>>(out, cyc, 10, 1e-6, gout);
  step_lab<V, FEAT><<<blocks, waves * 64, lds>>>(out, cyc, n_chunks, 1e-6, gout);
  (void)hipDeviceSynchronize();
  long long h[4];
  double ho[4];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
  std::printf("%-58s %d wave(s) x %d: %7.1f cycles per timestep   (checksum %.12g)\n", name, waves, blocks,
              static_cast<double>(h[0]) / (n_chunks * kChunkSteps), ho[0]);
  (void)hipFree(out);
  (void)hipFree(gout);
  (void)hipFree(cyc);
}

int main()
{
  run<16>("16: the kernel after round 2 (0 unguarded, krn in the flush)", 4, 1);
  run<15>("15: broadcast-free, one select + LDS store per step, unguarded", 4, 1);
  run<16>("16: the kernel after round 2 (0 unguarded, krn in the flush)", 1, 1);
  run<15>("15: broadcast-free, one select + LDS store per step, unguarded", 1, 1);
  for(int blocks : {1, 64, 256})
  {
    run<0>("0: the kernel's order", 4, blocks);
    run<12>("12: broadcast-free, gains -> HBM per chunk", 4, blocks);
    run<3>("3: Vxx recurrence only", 4, blocks);
  }
  run<0, 14>("0 without the per-step LDS operand loads", 1);
  run<0, 13>("0 without dV", 1);
  run<0, 11>("0 without the gains store", 1);
  run<0, 7>("0 without k_rel_norm", 1);
  run<0, 1>("0 without dV / gains / k_rel_norm", 1);
  run<0, 0>("0 without all four", 1);
  for(int waves : {1, 2, 4})
  {
    run<6, 14>("6 without the per-step LDS loads", waves);
    run<6, 11>("6 without the gain stores", waves);
    run<6, 10>("6 without loads and stores", waves);
    run<0, 14>("0 without the per-step LDS loads", waves);
  }
  for(int waves : {1, 4})
  {
    run<0>("0: the kernel's order", waves);
    run<3>("3: Vxx recurrence only (no dV / gains / krn)", waves);
    run<1>("1: Vx path split off and deferred, inside the MFMA phase", waves);
    run<4>("4: broadcast-free operands, Vx path after the chain", waves);
    run<5>("5: broadcast-free operands, Vx path deferred into the MFMAs", waves);
    run<6>("6: as 4, k_rel_norm moved out of the step", waves);
    run<7>("7: as 6, every lane its own dummy slot", waves);
    run<8>("8: single reads, krn in the step, gains -> HBM per chunk", waves);
    run<9>("9: as 8, gains -> LDS in a burst per chunk", waves);
    run<10>("10: as 8, Vx path deferred into the MFMAs", waves);
    run<11>("11: as 8 without the stores (gains summed up)", waves);
    run<12>("12: as 8 with mergeable loads (ds_read2_b64)", waves);
    run<13>("13: as 9 (LDS burst) with mergeable loads", waves);
    run<14>("14: as 12, operands of two timesteps requested together", waves);
    run<2>("2: Vx path split off and deferred, after the chain", waves);
  }
  return 0;
}
