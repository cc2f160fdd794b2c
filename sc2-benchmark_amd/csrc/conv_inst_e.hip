// Explicit instantiations of the implicit-GEMM launchers, group e (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace sc2conv {
template int launch8<B_gdn512>(const ConvArgs &, hipStream_t);
template int launch8<B_dec2>(const ConvArgs &, hipStream_t);
template int launch8<B_gdn256>(const ConvArgs &, hipStream_t);
template int launch8<B_dec4>(const ConvArgs &, hipStream_t);
}  // namespace sc2conv
