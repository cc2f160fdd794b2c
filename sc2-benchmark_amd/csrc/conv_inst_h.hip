// Explicit instantiations of the implicit-GEMM launchers, group h (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace sc2conv {
template int launch4<R_dec2>(const ConvArgs &, hipStream_t);
template int launch4<R_dec4>(const ConvArgs &, hipStream_t);
template int launch4<RG_256>(const ConvArgs &, hipStream_t);
template int launch8<S2_128>(const ConvArgs &, hipStream_t);
template int launch8<S2_256>(const ConvArgs &, hipStream_t);
}  // namespace sc2conv
