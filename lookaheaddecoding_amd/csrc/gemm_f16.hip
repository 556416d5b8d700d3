// f16: the skinny GEMM's dispatcher over the four groups of its shape table (each group is a translation unit of its own: gemm_part.hip)
#include "gemm_decl.hpp"

namespace lade {
int gemm_dispatch_f16_p0(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt);
int gemm_dispatch_f16_p1(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt);
int gemm_dispatch_f16_p2(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt);
int gemm_dispatch_f16_p3(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt);

int gemm_dispatch_f16(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt) {
    int rc = gemm_dispatch_f16_p0(g, st, mw, mt, ng, nt);
    if (rc == -1) rc = gemm_dispatch_f16_p1(g, st, mw, mt, ng, nt);
    if (rc == -1) rc = gemm_dispatch_f16_p2(g, st, mw, mt, ng, nt);
    if (rc == -1) rc = gemm_dispatch_f16_p3(g, st, mw, mt, ng, nt);
    return rc;
}
}  // namespace lade
