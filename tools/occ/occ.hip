// Diagnostic: resident workgroups per CU the runtime reports for the step's main kernels (hipOccupancyMaxActiveBlocksPerMultiprocessor).
#include "../../gdrf_amd/csrc/common.h"
#include "../../gdrf_amd/csrc/gemm_nt.h"
#include "../../gdrf_amd/csrc/gemm_tn.h"
#include "../../gdrf_amd/csrc/gemm_split.h"
#include "../../gdrf_amd/csrc/gemm_tn_topics.h"
#include "../../gdrf_amd/csrc/kernels_mm.h"
#include "../../gdrf_amd/csrc/kernels_n.h"
#include <cstdio>
using namespace gdrf;
template <class F> static void q(const char* name, F f, int threads, size_t lds) {
  int nb = -1;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)f, threads, lds);
  hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void*)f);
  printf("%-40s threads %4d dyn LDS %6zu static LDS %6zu regs %3d -> %d workgroups per CU (%s)\n", name, threads, lds, (size_t)a.sharedSizeBytes, a.numRegs, nb, hipGetErrorString(e));
}
int main() {
  using CS = NTCfg<double>;
  q("fwd_w  gemm_nt<double, FwdWProb>", gemm_nt_kernel<double, FwdWProb<double, float>>, 256, CS::LDS_BYTES);
  q("bwd_knm gemm_nt_v160<double, BwdKnmProb>", gemm_nt_kernel_v160<double, BwdKnmProb<double, float>>, 256, CS::LDS_BYTES);
  q("gemm_tn_split<SplitF16>", gemm_tn_split_kernel<SplitF16>, 256, 3 * 2 * 32 * 128 * 2);
  return 0;
}
