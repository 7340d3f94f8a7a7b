// K7, fp32-class, translation unit 2: dk = 128 with TWO key blocks per wave (round 5; see sparse_attn_x3p_impl.h) -- one wave per SIMD,
// 512 registers, launches of 5 .. 8 key blocks (129 .. 256 keys).
#include "sparse_attn_x3p_impl.h"

namespace snf {
namespace x3p {
int run_dk128_k2(const X3PParams& P, const X3PPlan& pl, float* out, hipStream_t s, int mode) {
#define SNF_X3P_CASE(NB) \
    case NB: return x3p_modes<128, NB, 2>(P, pl, out, s, mode);
    switch (pl.nkb) {
#ifndef SNF_ATTN_DEV
        SNF_X3P_CASE(5)
        SNF_X3P_CASE(6)
#endif
        SNF_X3P_CASE(7)
#ifndef SNF_ATTN_DEV
        SNF_X3P_CASE(8)
#endif
        default: break;
    }
#undef SNF_X3P_CASE
    snf::set_error("sparse_attn_x3p: key-block count %d not built (dk = 128, two key blocks per wave)", pl.nkb);
    return SNF_EUNSUPPORTED;
}
}  // namespace x3p
}  // namespace snf
