// f16 instantiations of the register-resident-activation GEMM (see gemm_ra.hpp)
#include "gemm_ra.hpp"

namespace lade {
int gemm_ra_dispatch_f16(const GemmRA& g, hipStream_t st, int mw, int cs) { return gemm_ra_dispatch<F16>(g, st, mw, cs); }
}  // namespace lade
