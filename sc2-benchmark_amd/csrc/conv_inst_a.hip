// Explicit instantiations of the implicit-GEMM launchers, group a (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace sc2conv {
template int launch<C_conv0>(const ConvArgs &, hipStream_t);
template int launch<C_gdn96>(const ConvArgs &, hipStream_t);
template int launch<C_conv2>(const ConvArgs &, hipStream_t);
template int launch<C_gdn48>(const ConvArgs &, hipStream_t);
template int launch<C_conv4>(const ConvArgs &, hipStream_t);
template int launch_patch<C_conv2>(const ConvArgs &, hipStream_t);
template int launch<Cx_gdn96>(const ConvArgs &, hipStream_t);
template int launch<Cx_gdn48>(const ConvArgs &, hipStream_t);
}  // namespace sc2conv
