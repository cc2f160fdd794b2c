// Explicit instantiations of the implicit-GEMM launchers, group f (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace sc2conv {
template int launch8<H_dec2>(const ConvArgs &, hipStream_t);
template int launch8<H_dec4>(const ConvArgs &, hipStream_t);
template int launch8<BG_256>(const ConvArgs &, hipStream_t);
template int launch8<BG_128>(const ConvArgs &, hipStream_t);
}  // namespace sc2conv
