// K7 (fast form), translation unit 2: the dk = 64 kernel variants (see sparse_attn_mfma_impl.h).
#include "sparse_attn_mfma_impl.h"

namespace snf {
int attn_launch_dk64(int qv_dtype, bool stats_pass, const AttnParams& P, const Plan& pl, float* out, hipStream_t s) {
    if (stats_pass)
        return qv_dtype == SNF_DT_F32 ? launch_stats<64, float>(P, pl, s) : launch_stats<64, unsigned short>(P, pl, s);
    return qv_dtype == SNF_DT_F32 ? launch_nkb<64, float>(P, pl, out, s) : launch_nkb<64, unsigned short>(P, pl, out, s);
}
}  // namespace snf
