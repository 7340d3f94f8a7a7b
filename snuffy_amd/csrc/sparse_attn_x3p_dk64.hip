// K7, fp32-class, translation unit 3: dk = 64 (round 5; see sparse_attn_x3p_impl.h) -- BASELINE config A and the D = 384 column of the
// north-star sweep; one key block per wave, launches of 4 .. 8 key blocks (97 .. 256 keys).
#include "sparse_attn_x3p_impl.h"

namespace snf {
namespace x3p {
int run_dk64_k1(const X3PParams& P, const X3PPlan& pl, float* out, hipStream_t s, int mode) {
#define SNF_X3P_CASE(NB) \
    case NB: return x3p_modes<64, NB, 1>(P, pl, out, s, mode);
    switch (pl.nkb) {
#ifndef SNF_ATTN_DEV
        SNF_X3P_CASE(4)
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
    snf::set_error("sparse_attn_x3p: key-block count %d not built (dk = 64)", pl.nkb);
    return SNF_EUNSUPPORTED;
}
}  // namespace x3p
}  // namespace snf
