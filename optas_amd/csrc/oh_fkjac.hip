// K1: batched forward kinematics + geometric Jacobian (+ reference-signed quaternion) of one root->link chain -- the kernel the north star
// names.  Replaces RobotModel.get_global_link_{position,quaternion,geometric_jacobian}_function(link, n=N)
// (optas/models.py:935-947,1090-1106,1199-1281).  One lane per unit, streaming: 56 B in, 56 + 336 B out per unit.
//   SOA=true : q[ndof][n], pose[7][n], J[6*ndof][n]      (coalesced; solver-internal / roofline)
//   SOA=false: q[n][ndof], pose[n][7], J[n][6][ndof]     (reference layout at the ABI)
// The kernel is templated on the chain length NC (1..8; NC = 0 takes it from the constants, up to OH_MAX_CHAIN): with the trip count
// known the per-joint axes and origins z[k], pj[k] are exactly NC x 3 registers and there is nothing to merge after skipped joints.  The
// round-1 kernel unrolled all 16 possible joints under run-time predicates: 254 VGPRs (2 waves per SIMD), ~1300 register moves and
// ~3000 executed instructions per unit for ~1000 of arithmetic -- it was VALU-issue bound at 43 % of the HBM roofline.
#include "oh_fkjac_unit.h"
#include "oh_kernels.h"

namespace {

template <bool SOA, int NC>
__global__ __launch_bounds__(256) void k_fk_jac(const oh_chain* __restrict__ ch, const int n, const double* __restrict__ q, double* __restrict__ pose,
                                                double* __restrict__ J) {
  fk_jac_unit<SOA, NC>(ch, n, q, pose, J);
}

template <bool SOA>
void launch(hipStream_t s, const oh_chain* d_chain, int nc, int ndof, int n, const double* q, double* pose, double* J) {
  const dim3 b(256), g((n + 255) / 256);
  const size_t lds = SOA ? 0 : oh_fk_tile_bytes(ndof);  // staging tile of the reference layout (oh_fkjac_unit.h)
  if (lds > 48 * 1024) {  // models with more than 8 joints: raise the dynamic-LDS ceiling of the run-time-length instantiation (up to 96 KB at 16)
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_fk_jac<SOA, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (nc >= 1 && nc <= 8) nc = 0;  // (the fixed-length instantiations are launched with at most 48 KB; such a chain takes the run-time-length one)
  }
  switch (nc) {
    case 1: hipLaunchKernelGGL((k_fk_jac<SOA, 1>), g, b, lds, s, d_chain, n, q, pose, J); break;
    case 2: hipLaunchKernelGGL((k_fk_jac<SOA, 2>), g, b, lds, s, d_chain, n, q, pose, J); break;
    case 3: hipLaunchKernelGGL((k_fk_jac<SOA, 3>), g, b, lds, s, d_chain, n, q, pose, J); break;
    case 4: hipLaunchKernelGGL((k_fk_jac<SOA, 4>), g, b, lds, s, d_chain, n, q, pose, J); break;
    case 5: hipLaunchKernelGGL((k_fk_jac<SOA, 5>), g, b, lds, s, d_chain, n, q, pose, J); break;
    case 6: hipLaunchKernelGGL((k_fk_jac<SOA, 6>), g, b, lds, s, d_chain, n, q, pose, J); break;
    case 7: hipLaunchKernelGGL((k_fk_jac<SOA, 7>), g, b, lds, s, d_chain, n, q, pose, J); break;
    case 8: hipLaunchKernelGGL((k_fk_jac<SOA, 8>), g, b, lds, s, d_chain, n, q, pose, J); break;
    default: hipLaunchKernelGGL((k_fk_jac<SOA, 0>), g, b, lds, s, d_chain, n, q, pose, J); break;
  }
}

}  // namespace

void oh_launch_fk_jac(hipStream_t s, bool soa, const oh_chain* d_chain, int n_chain, int ndof, int n, const double* q, double* pose, double* J) {
  if (soa) launch<true>(s, d_chain, n_chain, ndof, n, q, pose, J);
  else launch<false>(s, d_chain, n_chain, ndof, n, q, pose, J);
}

namespace {
template <class K>
bool kernel_info(K kernel, int block, OhKernelInfo* out) {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel)) != hipSuccess) return false;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess) nb = 0;
  *out = OhKernelInfo{a.numRegs, (int)a.localSizeBytes, (int)a.sharedSizeBytes, block, nb};
  return true;
}
}  // namespace
bool oh_kernel_info_fkjac(const char* name, OhKernelInfo* out) {
  if (std::string(name) == "k_fk_jac") return kernel_info(k_fk_jac<true, 7>, 256, out);
  return false;
}
