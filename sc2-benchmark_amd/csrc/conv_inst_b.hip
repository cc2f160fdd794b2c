// Explicit instantiations of the implicit-GEMM launchers, group b (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace sc2conv {
template int launch<C_dec0>(const ConvArgs &, hipStream_t);
template int launch<C_gdn512>(const ConvArgs &, hipStream_t);
template int launch<C_dec2>(const ConvArgs &, hipStream_t);
template int launch<C_gdn256>(const ConvArgs &, hipStream_t);
template int launch<C_dec4>(const ConvArgs &, hipStream_t);
template int launch<Cx_gdn512>(const ConvArgs &, hipStream_t);
template int launch<Cx_gdn256>(const ConvArgs &, hipStream_t);
}  // namespace sc2conv
