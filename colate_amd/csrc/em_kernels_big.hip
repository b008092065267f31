// colate_amd/csrc/em_kernels_big.hip -- the EM kernel (em_kernel_impl.hpp) for 257 .. 1024 epochs: 8 and 16 epochs per lane in the
// two-wave layout (the throughput variant's: one wave per role that walks through the bin groups).  The reference builds and runs any
// number of epochs (`--bins 3,7,0.01` gives 404, include/coal/coal.cpp:3551-3632); these instantiations exist so that the
// drop-in does too.  Same template, same arithmetic; the per-epoch arrays no longer fit the register file (the compiler keeps part
// of them in scratch memory) and only the general loop is compiled: speed is secondary here.
#include "em_kernel_impl.hpp"

hipError_t colate_em_launch_big(const ColateEmArgs& args, hipStream_t stream) {
  const int nch = em_chunks(args.E);
  if (args.mode == 1) {
    const int threads = em_threads(args.A);
    const size_t lds = em_lds_bytes(args.E, args.A, false);
    if (nch == 8) return launch_one<1, 8, 4, false>(args, stream, lds, threads);
    return launch_one<1, 16, 4, false>(args, stream, lds, threads);
  }
  const size_t lds = em_lds_bytes(args.E, args.A, true);
  if (nch == 8) return launch_one<0, 8, 4, true>(args, stream, lds, 2 * kWave);
  return launch_one<0, 16, 4, true>(args, stream, lds, 2 * kWave);
}
