// ISA lab (r5): one or two instantiations of the ring kernel, compiled alone so that the inner loop's schedule can be read with
// llvm-objdump in seconds (conv.hip instantiates ~70 and takes two minutes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -I include -I lsfa_amd/csrc \
//         -c tools/lab/ring_isa.hip -o tools/lab/_build/ring_isa.o
#include "conv_ring_kernel.h"
using namespace lsfa::convsplit;
#ifndef ISA_NT
#define ISA_NT 4
#endif
#ifndef ISA_PC
#define ISA_PC 2
#endif
#ifndef ISA_ST
#define ISA_ST 3
#endif
#ifndef ISA_SP
#define ISA_SP true
#endif
#ifndef ISA_AF
#define ISA_AF false
#endif
#ifndef ISA_WV
#define ISA_WV 4
#endif
template __global__ void lsfa::convsplit::conv_ring_kernel<ISA_NT, ISA_PC, ISA_ST, ISA_SP, ISA_AF, ISA_WV>(Args, int, int, int);
