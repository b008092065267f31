// colate_amd/csrc/em_kernels_ilp.hip -- the latency variant of the EM kernel (em_kernel_impl.hpp) built with the
// `max-ilp` machine-scheduler strategy (see Makefile).  That strategy orders the long dependent FP64 chains of an
// iteration for instruction-level parallelism at the price of ~8 more VGPRs, which takes the kernel from 3 to 2 waves
// per SIMD: worth 2.7 % when every workgroup has a CU to itself (B <= #CUs: all BASELINE configurations), harmful
// when workgroups must share CUs.  colate_em_launch (em_kernels.hip) therefore calls into this unit only for
// B <= #CUs and E <= 128; everything else runs the default-scheduler build of the same template.
#define COLATE_EM_ILP_BUILD 1
#include "em_kernel_impl.hpp"

hipError_t colate_em_launch_latency_ilp(const ColateEmArgs& args, hipStream_t stream) { return launch_latency(args, stream); }
