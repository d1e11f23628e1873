// inst_scan_mfma.hip -- every instantiation of the matrix-core scan kernel the host launches (launch_scan_mfma_* in rabitq_hip.hip),
// compiled in a translation unit of its own: an edit of the kernel recompiles this object only.
#include "common.h"
#include "kernels_scan_mfma.h"

#define RQ_MFMA_PARAMS                                                                                                               \
    const uint32_t *, const float4 *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, SurvRec *, RunRec *,   \
        unsigned long long *, unsigned long long *, const uint4 *, const float4 *, const float4 *, const ScanArgs
#define RQ_INST(W, NT)                                                                                                               \
    template __global__ void scan_mfma_kernel<W, NT, false, false>(RQ_MFMA_PARAMS);                                                  \
    template __global__ void scan_mfma_kernel<W, NT, true, false>(RQ_MFMA_PARAMS);
RQ_INST(1, 4)
RQ_INST(2, RQ_NT_W2)
RQ_INST(3, 4)
RQ_INST(4, 2)
RQ_INST(6, 2)
RQ_INST(8, 2)
RQ_INST(12, RQ_NT_W12)
RQ_INST(16, 2)
#undef RQ_INST
// the additive-gate instantiations (dim 64 / 128, uniform survivor buffers)
template __global__ void scan_mfma_kernel<1, 4, false, true>(RQ_MFMA_PARAMS);
template __global__ void scan_mfma_kernel<2, RQ_ADD_NT2, false, true>(RQ_MFMA_PARAMS);
