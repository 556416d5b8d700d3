// f16 instantiations of the ping-pong form of the skinny GEMM (see gemm_pp.hpp)
#include "gemm_pp.hpp"

namespace lade {
int gemm_pp_dispatch_f16(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt) { return gemm_pp_dispatch<F16>(g, st, mw, mt, ng, nt); }
}  // namespace lade
