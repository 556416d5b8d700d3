// bf16 instantiations of the register-resident-activation GEMM (see gemm_ra.hpp)
#include "gemm_ra.hpp"

namespace lade {
int gemm_ra_dispatch_bf16(const GemmRA& g, hipStream_t st, int mw, int cs) { return gemm_ra_dispatch<BF16>(g, st, mw, cs); }
}  // namespace lade
