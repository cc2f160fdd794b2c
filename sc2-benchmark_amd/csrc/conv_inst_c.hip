// Explicit instantiations of the implicit-GEMM launchers, group c (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace sc2conv {
template int launch<G_128>(const ConvArgs &, hipStream_t);
template int launch<G_96>(const ConvArgs &, hipStream_t);
template int launch<G_64>(const ConvArgs &, hipStream_t);
template int launch<G_48>(const ConvArgs &, hipStream_t);
template int launch<G_32>(const ConvArgs &, hipStream_t);
template int launch<Gd_128>(const ConvArgs &, hipStream_t);
}  // namespace sc2conv
