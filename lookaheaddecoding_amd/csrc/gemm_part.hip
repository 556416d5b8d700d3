// One group of the skinny GEMM's shape table for one dtype (gemm_kernel.hpp: GO_0 .. GO_3), compiled eight times by the Makefile:
//   -DLADE_GEMM_T=BF16|F16 -DLADE_GEMM_TN=bf16|f16 -DLADE_GEMM_PART=0..3
#include "gemm_kernel.hpp"

#define LADE_CAT_(a, b, c) a##b##_p##c
#define LADE_CAT(a, b, c) LADE_CAT_(a, b, c)

namespace lade {
int LADE_CAT(gemm_dispatch_, LADE_GEMM_TN, LADE_GEMM_PART)(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt) {
    return gemm_dispatch_part<LADE_GEMM_T, LADE_GEMM_PART>(g, st, mw, mt, ng, nt);
}
}  // namespace lade
