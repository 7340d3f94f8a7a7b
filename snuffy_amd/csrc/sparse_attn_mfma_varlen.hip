// K7 (fast form), translation unit 3: the varlen (many bags per launch) variants, dk = 128 (see sparse_attn_mfma_impl.h).
#include "sparse_attn_mfma_impl.h"

namespace snf {
int attn_launch_varlen_dk128(const AttnParams& P, const Plan& pl, float* out, hipStream_t s) {
    return launch_nkb_varlen<128>(P, pl, out, s);
}
}  // namespace snf
