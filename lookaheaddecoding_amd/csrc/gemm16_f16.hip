// f16 instantiations of the 16-row-granular skinny GEMM (gemm16.hpp)
#include "gemm16.hpp"

namespace lade {
int gemm16_dispatch_f16(const GemmK& g, hipStream_t st, int mt16, int ng, int nt16) { return gemm16_dispatch<F16>(g, st, mt16, ng, nt16); }
}  // namespace lade
