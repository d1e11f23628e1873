// inst_scan_valu.hip -- every instantiation of the VALU scan kernel the host launches (launch_scan_t in rabitq_hip.hip), compiled in a
// translation unit of its own: an edit of the kernel recompiles this object only.
#include "common.h"
#include "kernels_scan_valu.h"

#define RQ_INST(W, CPL)                                                                                                              \
    template __global__ void scan_kernel<W, CPL, false>(SCAN_PARAMS);                                                                \
    template __global__ void scan_kernel<W, CPL, true>(SCAN_PARAMS);
RQ_INST(1, 2)
RQ_INST(2, 2)
RQ_INST(3, 2)
RQ_INST(4, 2)
RQ_INST(6, 2)
RQ_INST(8, 2)
RQ_INST(12, 1)
RQ_INST(16, 1)
#undef RQ_INST
