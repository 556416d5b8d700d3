// f16 instantiations of the skinny GEMM (see gemm_kernel.hpp)
#include "gemm_kernel.hpp"

namespace lade {
int gemm_dispatch_f16(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt) { return gemm_dispatch<F16>(g, st, mw, mt, ng, nt); }
}  // namespace lade
