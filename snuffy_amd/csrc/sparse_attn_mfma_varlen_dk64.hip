// K7 (fast form), translation unit 4: the varlen (many bags per launch) variants, dk = 64 (see sparse_attn_mfma_impl.h).
#include "sparse_attn_mfma_impl.h"

namespace snf {
int attn_launch_varlen_dk64(const AttnParams& P, const Plan& pl, float* out, hipStream_t s) {
    return launch_nkb_varlen<64>(P, pl, out, s);
}
}  // namespace snf
