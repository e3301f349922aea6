// Body of K1 (batched forward kinematics + geometric Jacobian of one chain, see oh_fkjac.hip), one lane per unit.  oh_fkjac.hip wraps it in
// the generic kernels (chain through a device pointer); oh_jit.hip compiles it behind a constexpr copy of a handle's chain.
#pragma once
#include "oh_device.h"

// dynamic LDS of the reference-layout kernel (blocks of 256): J of 128 units at a time, and room for q and pose of all 256
inline size_t oh_fk_tile_bytes(const int ndof) { return sizeof(double) * (size_t)(768 * ndof > 1792 ? 768 * ndof : 1792); }

template <bool SOA, int NC>
__device__ void fk_jac_unit(const oh_chain* __restrict__ ch, const int n, const double* __restrict__ q, double* __restrict__ pose, double* __restrict__ J) {
  constexpr int NM = NC ? NC : OH_MAX_CHAIN;
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
  const bool alive = u < (unsigned)n;
  if (SOA && !alive) return;  // (the reference layout goes through block-wide LDS staging: every lane stays for the barriers)
  const int nc = NC ? NC : ch->n_chain;
  const int ndof = ch->ndof;
  const bool want_pose = pose != nullptr;
  // Reference layout (q[n][ndof], pose[n][7], J[n][6][ndof]): a lane's values are 56 ... 336 B apart from its neighbour's, so lane-wise
  // loads and stores touch 64 different lines per instruction (round 2 measured 2.0 TB/s against 5.1 in the SoA layout).  The rows of
  // the block's units are contiguous in memory, though: they are staged in LDS, lane-wise on one side and linearly -- full 512-byte
  // lines per wave instruction -- on the memory side.  Tile: 128 units x 6 ndof doubles (J of half the block at a time; q and pose fit).
  extern __shared__ double oh_fk_tile[];
  const unsigned u0 = blockIdx.x * blockDim.x;
  const unsigned nblk = (unsigned)n - u0 < blockDim.x ? (unsigned)n - u0 : blockDim.x;  // live units of this block
  auto block_copy_out = [&](double* __restrict__ dst, const unsigned count) {  // LDS tile[0 .. count) -> memory, linear
    for (unsigned i = threadIdx.x; i < count; i += blockDim.x) dst[i] = oh_fk_tile[i];
  };
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double p[3] = {0, 0, 0};
  double quat[4] = {0, 0, 0, 1};
  double z[NM][3], pj[NM][3];
  double qk[NM];
  if constexpr (!SOA) {
    const double* __restrict__ src = q + (size_t)u0 * ndof;
    for (unsigned i = threadIdx.x; i < nblk * (unsigned)ndof; i += blockDim.x) oh_fk_tile[i] = src[i];
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NM; ++k) {
    if (NC || k < nc) {
      const int qi = ch->qidx[k];
      if constexpr (SOA) qk[k] = (q + (size_t)qi * n)[u];  // uniform base, 32-bit lane offset
      else qk[k] = alive ? oh_fk_tile[threadIdx.x * ndof + qi] : 0.0;
    }
  }
  if constexpr (!SOA) __syncthreads();  // the tile is reused for the outputs
#pragma unroll
  for (int k = 0; k < NM; ++k) {
    if (NC || k < nc) {
      double t[3];
      mv3(R, ch->p0[k], t);
      p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
      if (!ch->r0ident[k]) {
        double Rn[9];
        mm3(R, ch->R0[k], Rn);
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
      }
      pj[k][0] = p[0]; pj[k][1] = p[1]; pj[k][2] = p[2];
      if (ch->jtype[k] == 0) {
        double sh, chh;
        sincos_joint(0.5 * qk[k], &sh, &chh);  // half angle: quaternion (spatialmath.py:372-375) ...
        const double s = 2.0 * sh * chh, c = 1.0 - 2.0 * sh * sh;  // ... and full angle for Rodrigues
        if (ch->axcode[k] != 0) rot_principal_right(R, ch->axcode[k], s, c, z[k]);
        else rot_axis_right(R, ch->axis[k], s, c, z[k]);
        if (want_pose) {
          double qn[4];
          qmul(quat, ch->quat0[k], qn);  // == fromrpy(rpy) * quat in the reference's reversed product
          const double qa[4] = {sh * ch->axis[k][0], sh * ch->axis[k][1], sh * ch->axis[k][2], chh};
          qmul(qn, qa, quat);
        }
      } else {
        mv3(R, ch->axis[k], z[k]);
        p[0] += z[k][0] * qk[k]; p[1] += z[k][1] * qk[k]; p[2] += z[k][2] * qk[k];
        if (want_pose) {
          double qn[4];
          qmul(quat, ch->quat0[k], qn);
          quat[0] = qn[0]; quat[1] = qn[1]; quat[2] = qn[2]; quat[3] = qn[3];
        }
      }
    }
  }
  double e[3], t[3];
  mv3(R, ch->p_tool, t);
  e[0] = p[0] + t[0]; e[1] = p[1] + t[1]; e[2] = p[2] + t[2];
  if (want_pose) {
    double qe[4];
    qmul(quat, ch->quat_tool, qe);
    const double o[7] = {e[0], e[1], e[2], qe[0], qe[1], qe[2], qe[3]};
    if constexpr (SOA) {
#pragma unroll
      for (int i = 0; i < 7; ++i) (pose + (size_t)i * n)[u] = o[i];
    } else {
#pragma unroll
      for (int i = 0; i < 7; ++i) oh_fk_tile[threadIdx.x * 7 + i] = o[i];
      __syncthreads();
      block_copy_out(pose + (size_t)u0 * 7, nblk * 7u);
      __syncthreads();
    }
  }
  if (J) {
    // columns of joints that are not on the chain are zero (models.py:1251-1254)
    if (SOA && nc != ndof) {
      for (int i = 0; i < 6 * ndof; ++i) (J + (size_t)i * n)[u] = 0.0;
    }
    const unsigned half = blockDim.x / 2;
    const int rowlen = 6 * ndof;
    double* __restrict__ mine = oh_fk_tile + (size_t)(threadIdx.x % half) * rowlen;  // this lane's row of the tile in its pass
    for (unsigned pass = 0; pass < (SOA ? 1u : 2u); ++pass) {
    const bool stage = !SOA && (threadIdx.x / half) == pass;
    if (stage && nc != ndof)
      for (int i = 0; i < rowlen; ++i) mine[i] = 0.0;
#pragma unroll
    for (int k = 0; k < NM; ++k) {
      if (NC || k < nc) {
        const int col = ch->qidx[k];
        double col6[6];
        if (ch->jtype[k] == 0) {
          const double d[3] = {e[0] - pj[k][0], e[1] - pj[k][1], e[2] - pj[k][2]};
          cross3(z[k], d, col6);  // models.py:1236-1239
          col6[3] = z[k][0]; col6[4] = z[k][1]; col6[5] = z[k][2];
        } else {
          col6[0] = z[k][0]; col6[1] = z[k][1]; col6[2] = z[k][2];  // models.py:1245-1246
          col6[3] = col6[4] = col6[5] = 0.0;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          if constexpr (SOA) (J + (size_t)(r * ndof + col) * n)[u] = col6[r];
          else if (stage) mine[r * ndof + col] = col6[r];
        }
      }
    }
    if constexpr (!SOA) {
      __syncthreads();
      const unsigned first = pass * half;  // units of this pass: [first, first + half) of the block
      if (nblk > first) block_copy_out(J + (size_t)(u0 + first) * rowlen, ((nblk - first < half) ? nblk - first : half) * (unsigned)rowlen);
      __syncthreads();
    }
    }
  }
}
