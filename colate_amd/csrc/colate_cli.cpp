// colate_amd/csrc/colate_cli.cpp -- the `Colate` executable of colate_amd: the
// reference's command line for `--mode mut` (include/coal/Colate.cpp:6-116),
// implemented in libcolate_amd.so (colate_mut_main).
#include "colate_amd.h"

int main(int argc, char** argv) { return colate_mut_main(argc, argv); }
