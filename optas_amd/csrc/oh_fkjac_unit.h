// Body of K1 (batched forward kinematics + geometric Jacobian of one chain, see oh_fkjac.hip), one lane per unit.  oh_fkjac.hip wraps it in
// the generic kernels (chain through a device pointer); oh_jit.hip compiles it behind a constexpr copy of a handle's chain.
#pragma once
#include "oh_device.h"

template <bool SOA, int NC>
__device__ void fk_jac_unit(const oh_chain* __restrict__ ch, const int n, const double* __restrict__ q, double* __restrict__ pose, double* __restrict__ J) {
  constexpr int NM = NC ? NC : OH_MAX_CHAIN;
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (unsigned)n) return;
  const int nc = NC ? NC : ch->n_chain;
  const int ndof = ch->ndof;
  const bool want_pose = pose != nullptr;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double p[3] = {0, 0, 0};
  double quat[4] = {0, 0, 0, 1};
  double z[NM][3], pj[NM][3];
  double qk[NM];
#pragma unroll
  for (int k = 0; k < NM; ++k) {
    if (NC || k < nc) {
      const int qi = ch->qidx[k];
      const double* row = SOA ? q + (size_t)qi * n : q + qi;  // uniform base, 32-bit lane offset
      qk[k] = SOA ? row[u] : row[(size_t)u * ndof];
    }
  }
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
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      if (SOA) (pose + (size_t)i * n)[u] = o[i];
      else pose[(size_t)u * 7 + i] = o[i];
    }
  }
  if (J) {
    // columns of joints that are not on the chain are zero (models.py:1251-1254)
    if (nc != ndof) {
      for (int i = 0; i < 6 * ndof; ++i) {
        if (SOA) (J + (size_t)i * n)[u] = 0.0;
        else J[(size_t)u * 6 * ndof + i] = 0.0;
      }
    }
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
          if (SOA) (J + (size_t)(r * ndof + col) * n)[u] = col6[r];
          else J[(size_t)u * 6 * ndof + r * ndof + col] = col6[r];
        }
      }
    }
  }
}
