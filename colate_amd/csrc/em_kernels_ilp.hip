// colate_amd/csrc/em_kernels_ilp.hip -- the latency variant of the EM kernel (em_kernel_impl.hpp) built with the
// `max-ilp` machine-scheduler strategy (see Makefile).  That strategy orders the long dependent FP64 chains of an
// iteration for instruction-level parallelism at the price of a few more VGPRs; since the per-kind loops it still
// fits 3 waves per SIMD at E <= 64 and is the faster latency build at every batch size, so colate_em_variant
// (em_kernels.hip) picks it for every latency launch: B <= 2 x #CUs, E <= 128.
#define COLATE_EM_ILP_BUILD 1
#include "em_kernel_impl.hpp"

hipError_t colate_em_launch_latency_ilp(const ColateEmArgs& args, hipStream_t stream, bool alone) { return launch_latency(args, stream, alone); }
