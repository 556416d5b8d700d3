// bf16 instantiations of the skinny GEMM (see gemm_kernel.hpp)
#include "gemm_kernel.hpp"

namespace lade {
int gemm_dispatch_bf16(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt) { return gemm_dispatch<BF16>(g, st, mw, mt, ng, nt); }
}  // namespace lade
