// Explicit instantiations of the implicit-GEMM launchers, group g (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace sc2conv {
template int launch8<P3_256>(const ConvArgs &, hipStream_t);
template int launch8<P3_128>(const ConvArgs &, hipStream_t);
template int launch8<P2_dec2>(const ConvArgs &, hipStream_t);
template int launch8<P2_dec4>(const ConvArgs &, hipStream_t);
}  // namespace sc2conv
