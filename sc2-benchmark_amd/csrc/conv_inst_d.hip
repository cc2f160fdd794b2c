// Explicit instantiations of the implicit-GEMM launchers, group d (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace sc2conv {
template int launch<Gx_128>(const ConvArgs &, hipStream_t);
template int launch<Gx_96>(const ConvArgs &, hipStream_t);
template int launch<Gx_64>(const ConvArgs &, hipStream_t);
template int launch<Gx_48>(const ConvArgs &, hipStream_t);
template int launch<Gx_32>(const ConvArgs &, hipStream_t);
template int launch<Gq_128>(const ConvArgs &, hipStream_t);
template int launch<Gqx_128>(const ConvArgs &, hipStream_t);
template int launch<Gb_128>(const ConvArgs &, hipStream_t);
template int launch<Gb_96>(const ConvArgs &, hipStream_t);
}  // namespace sc2conv
