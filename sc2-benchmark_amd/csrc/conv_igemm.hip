// C-ABI entry points of the implicit-GEMM convolution family (kernels: conv_igemm_impl.h, instantiated in conv_inst_*.hip).
#include "conv_igemm_impl.h"

namespace sc2conv {
extern template int launch<C_conv0>(const ConvArgs &, hipStream_t);
extern template int launch<C_gdn96>(const ConvArgs &, hipStream_t);
extern template int launch<C_conv2>(const ConvArgs &, hipStream_t);
extern template int launch<C_gdn48>(const ConvArgs &, hipStream_t);
extern template int launch<C_conv4>(const ConvArgs &, hipStream_t);
extern template int launch_patch<C_conv2>(const ConvArgs &, hipStream_t);
extern template int launch<Cx_gdn96>(const ConvArgs &, hipStream_t);
extern template int launch<Cx_gdn48>(const ConvArgs &, hipStream_t);
extern template int launch<C_dec0>(const ConvArgs &, hipStream_t);
extern template int launch<C_gdn512>(const ConvArgs &, hipStream_t);
extern template int launch<C_dec2>(const ConvArgs &, hipStream_t);
extern template int launch<C_gdn256>(const ConvArgs &, hipStream_t);
extern template int launch<C_dec4>(const ConvArgs &, hipStream_t);
extern template int launch<Cx_gdn512>(const ConvArgs &, hipStream_t);
extern template int launch<Cx_gdn256>(const ConvArgs &, hipStream_t);
extern template int launch<G_128>(const ConvArgs &, hipStream_t);
extern template int launch<G_96>(const ConvArgs &, hipStream_t);
extern template int launch<G_64>(const ConvArgs &, hipStream_t);
extern template int launch<G_48>(const ConvArgs &, hipStream_t);
extern template int launch<G_32>(const ConvArgs &, hipStream_t);
extern template int launch<Gd_128>(const ConvArgs &, hipStream_t);
extern template int launch<Gb_128>(const ConvArgs &, hipStream_t);
extern template int launch<Gb_96>(const ConvArgs &, hipStream_t);
extern template int launch<Gx_128>(const ConvArgs &, hipStream_t);
extern template int launch<Gx_96>(const ConvArgs &, hipStream_t);
extern template int launch<Gx_64>(const ConvArgs &, hipStream_t);
extern template int launch<Gx_48>(const ConvArgs &, hipStream_t);
extern template int launch<Gx_32>(const ConvArgs &, hipStream_t);
extern template int launch<Gq_128>(const ConvArgs &, hipStream_t);
extern template int launch<Gqx_128>(const ConvArgs &, hipStream_t);
extern template int launch8<B_gdn512>(const ConvArgs &, hipStream_t);
extern template int launch8<B_dec2>(const ConvArgs &, hipStream_t);
extern template int launch8<B_gdn256>(const ConvArgs &, hipStream_t);
extern template int launch8<B_dec4>(const ConvArgs &, hipStream_t);
extern template int launch8<H_dec2>(const ConvArgs &, hipStream_t);
extern template int launch8<H_dec4>(const ConvArgs &, hipStream_t);
extern template int launch8<BG_256>(const ConvArgs &, hipStream_t);
extern template int launch8<BG_128>(const ConvArgs &, hipStream_t);
extern template int launch8<P3_256>(const ConvArgs &, hipStream_t);
extern template int launch8<P3_128>(const ConvArgs &, hipStream_t);
extern template int launch8<S2_128>(const ConvArgs &, hipStream_t);
extern template int launch8<S2_256>(const ConvArgs &, hipStream_t);
extern template int launch8<P2_dec2>(const ConvArgs &, hipStream_t);
extern template int launch8<P2_dec4>(const ConvArgs &, hipStream_t);
extern template int launch4<R_dec2>(const ConvArgs &, hipStream_t);
extern template int launch4<R_dec4>(const ConvArgs &, hipStream_t);
extern template int launch4<RG_256>(const ConvArgs &, hipStream_t);
template <class C, int MODE> int launch8p(const ConvArgs &, hipStream_t);   // conv_dec_persist.hip
extern template int launch8p<B_dec2, 0>(const ConvArgs &, hipStream_t);
extern template int launch8p<B_dec4, 0>(const ConvArgs &, hipStream_t);
extern template int launch8p<B_dec2, 1>(const ConvArgs &, hipStream_t);
extern template int launch8p<B_dec4, 1>(const ConvArgs &, hipStream_t);
extern template int launch8p<B_dec2, 2>(const ConvArgs &, hipStream_t);
extern template int launch8p<B_dec4, 2>(const ConvArgs &, hipStream_t);
extern template int launch8p<B_dec2, 3>(const ConvArgs &, hipStream_t);
extern template int launch8p<B_dec4, 3>(const ConvArgs &, hipStream_t);
}  // namespace sc2conv
using namespace sc2conv;

extern "C" int sc2_conv_weight_rows(int Cout) {
    if (Cout <= 0) return 0;
    if (Cout <= 32) return 32;
    if (Cout <= 48) return 48;
    if (Cout <= 64) return 64;
    if (Cout <= 96) return 96;
    return (Cout + 127) / 128 * 128;
}
extern "C" int sc2_conv_weight_pitch(int K) { return K <= 0 ? 0 : (K + 63) / 64 * 64; }

namespace {
// the 8-wave 256-row tile is used when ...
bool big_tile_eligible(const sc2_conv_desc *d, long long M, int K) {
    // Measured (tools/attic/ab_big.py, one process, MI355X): 256-wide big tile wins +27 % at K = 2048 and +13 % at K = 1024,
    // ties or loses on the HBM-bound 1x1 GDN GEMMs and on short K; the 128-wide big tile never wins.
    // (SC2_CONV_FORCE_BIG / SC2_CONV_NO_BIG: test and A/B switches)
    const bool forced = sc2_pol().conv_force_big != 0;
    return sc2_conv_weight_rows(d->Cout) % 128 == 0 && !sc2_pol().conv_no_big &&
           ((K >= 1024 && d->Cout % 256 == 0 && d->a_op == SC2_AOP_NONE && M >= 256LL * 192) || forced);
}
}  // namespace

extern "C" int sc2_conv_patch_supported(const sc2_conv_desc *d) {
    if (!d) return 0;
    return d->Cin == 96 && d->Cout == 48 && d->KH == 5 && d->KW == 5 && d->stride_h == 2 && d->stride_w == 2 &&
                   d->pad_h == 2 && d->pad_w == 2 && d->OW > 0 && d->OW <= 64 && d->out_format == SC2_OUT_BF16_NHWC &&
                   d->out_H == 0 && d->a_op == SC2_AOP_NONE && !epi_needs_x(d->epilogue)
               ? 1 : 0;
}

extern "C" int sc2_conv_fused_gdn_supported(const sc2_conv_desc *d) {
    if (!d || d->Cout <= 0) return 0;
    if (d->Cout == 32 || d->Cout == 48 || d->Cout == 64 || d->Cout == 96) return 1;
    const long long M = (long long)d->N * d->OH * d->OW;
    return d->Cout == 256 && d->out_format != SC2_OUT_F32_NCHW && d->out_H == 0 &&
           big_tile_eligible(d, M, d->KH * d->KW * d->Cin) ? 2 : 0;
}

namespace {
int conv2d_fwd_impl(const sc2_conv_desc *d, const void *x, const void *w_packed, void *y, const void *ep_x, const float *ep_beta,
                    void *stream, const void *ep_x2, void *y2);
}
extern "C" int sc2_conv2d_fwd(const sc2_conv_desc *d, const void *x, const void *w_packed, void *y, const void *ep_x,
                              const float *ep_beta, void *stream) {
    SC2_REQUIRE(d && d->epilogue <= SC2_EPI_IGDN2, SC2_ERR_INVALID_ARG, "conv2d: bad epilogue (the GDN1-backward forms go through sc2_gdn1_bwd_gemm)");
    return conv2d_fwd_impl(d, x, w_packed, y, ep_x, ep_beta, stream, nullptr, nullptr);
}
extern "C" int sc2_gdn1_bwd_gemm(const sc2_conv_desc *d, const void *x, const void *w_packed, void *y, void *y2, const void *ep_x,
                                 const void *ep_x2, const float *ep_beta, void *stream) {
    SC2_REQUIRE(d && d->epilogue >= SC2_EPI_GDN1_BWD_PRE && d->epilogue <= SC2_EPI_GDN1_BWD_POST, SC2_ERR_INVALID_ARG,
                "gdn1_bwd_gemm: epilogue must be one of SC2_EPI_GDN1_BWD_PRE / IGDN1_BWD_PRE / GDN1_BWD_POST");
    const bool post = d->epilogue == SC2_EPI_GDN1_BWD_POST;
    SC2_REQUIRE(ep_x && ep_x2 && (post || (ep_beta && y2)), SC2_ERR_INVALID_ARG, "gdn1_bwd_gemm: null operand");
    SC2_REQUIRE(d->KH == 1 && d->KW == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0 && d->Cin == d->Cout &&
                    d->out_format == SC2_OUT_BF16_NHWC && d->out_H == 0 && d->dil_h <= 1 && d->dil_w <= 1 &&
                    !(d->k_order & SC2_K_B_FRAG_MAJOR) && (d->Cout_pad % 128 == 0 || d->Cout_pad == 96) &&
                    d->a_op == (post ? SC2_AOP_NONE : SC2_AOP_ABS),
                SC2_ERR_UNSUPPORTED, "gdn1_bwd_gemm: a 1x1 C x C GEMM on bf16 NHWC with 96 or a multiple of 128 packed rows (got %d -> %d, %d rows)",
                d->Cin, d->Cout, d->Cout_pad);
    static const float one = 1.0f;    // (the shared argument checks want a non-null ep_beta for any epilogue; POST reads none)
    return conv2d_fwd_impl(d, x, w_packed, y, ep_x, post ? &one : ep_beta, stream, ep_x2, y2);
}
namespace {
int conv2d_fwd_impl(const sc2_conv_desc *d, const void *x, const void *w_packed, void *y, const void *ep_x, const float *ep_beta,
                    void *stream, const void *ep_x2, void *y2) {
    SC2_REQUIRE(d && x && w_packed && y, SC2_ERR_INVALID_ARG, "conv2d: null argument");
    SC2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, SC2_ERR_INVALID_ARG,
                "conv2d: non-positive dimension");
    SC2_REQUIRE(d->Cin % 8 == 0 && d->Cout % 8 == 0, SC2_ERR_INVALID_ARG,
                "conv2d: Cin (%d) and Cout (%d) must be multiples of 8", d->Cin, d->Cout);
    SC2_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
                SC2_ERR_INVALID_ARG, "conv2d: bad filter geometry");
    SC2_REQUIRE(d->dil_h >= 0 && d->dil_w >= 0, SC2_ERR_INVALID_ARG, "conv2d: negative dilation");
    const int dil_h = d->dil_h > 0 ? d->dil_h : 1, dil_w = d->dil_w > 0 ? d->dil_w : 1;   // (0 = 1: descriptors written before the field existed)
    const bool dilated = dil_h != 1 || dil_w != 1;
    int OH = (d->H + 2 * d->pad_h - dil_h * (d->KH - 1) - 1) / d->stride_h + 1;
    int OW = (d->W + 2 * d->pad_w - dil_w * (d->KW - 1) - 1) / d->stride_w + 1;
    const bool scatter = d->out_H > 0;
    if (scatter) {
        // transposed-convolution use: the caller fixes the number of output rows/cols (rows past the symmetric
        // formula see implicit zero padding) and where each lands in a strided NHWC output
        SC2_REQUIRE(d->OH > 0 && d->OW > 0 && d->out_W > 0 && d->out_stride_h > 0 && d->out_stride_w > 0 &&
                        d->out_off_h >= 0 && d->out_off_w >= 0 && d->out_format != SC2_OUT_F32_NCHW &&
                        d->out_format != SC2_OUT_I32_NCHW_SYM,
                    SC2_ERR_INVALID_ARG, "conv2d: bad output scatter");
        OH = d->OH;
        OW = d->OW;
    }
    SC2_REQUIRE(OH == d->OH && OW == d->OW && OH > 0 && OW > 0, SC2_ERR_INVALID_ARG,
                "conv2d: output size %dx%d does not match geometry (%dx%d)", d->OH, d->OW, OH, OW);
    const int K = d->KH * d->KW * d->Cin;
    SC2_REQUIRE(d->Kpad == sc2_conv_weight_pitch(K), SC2_ERR_INVALID_ARG, "conv2d: Kpad %d != %d", d->Kpad,
                sc2_conv_weight_pitch(K));
    SC2_REQUIRE(d->Cout_pad == sc2_conv_weight_rows(d->Cout), SC2_ERR_INVALID_ARG, "conv2d: Cout_pad %d != %d",
                d->Cout_pad, sc2_conv_weight_rows(d->Cout));
    SC2_REQUIRE(d->a_op >= SC2_AOP_NONE && d->a_op <= SC2_AOP_SQUARE, SC2_ERR_INVALID_ARG, "conv2d: bad a_op");
    SC2_REQUIRE(d->k_order >= 0 && d->k_order <= 7 && (!(d->k_order & SC2_K_SLAB_MAJOR) || d->Cin % 32 == 0) &&
                    (!(d->k_order & SC2_K_B_FRAG_MAJOR) ||
                     ((d->k_order & SC2_K_SLAB_MAJOR) && !(d->k_order & SC2_K_B_TILE_MAJOR))),
                SC2_ERR_INVALID_ARG, "conv2d: slab-major K order needs Cin %% 32 == 0 (Cin = %d)", d->Cin);
    SC2_REQUIRE(d->epilogue >= SC2_EPI_NONE && d->epilogue <= SC2_EPI_GDN1_BWD_POST, SC2_ERR_INVALID_ARG,
                "conv2d: bad epilogue");
    const bool gdn_bwd = d->epilogue >= SC2_EPI_GDN1_BWD_PRE;
    const bool fused = d->epilogue == SC2_EPI_FUSED_GDN || d->epilogue == SC2_EPI_FUSED_IGDN;
    if (fused)
        SC2_REQUIRE(d->Cout == d->Cout_pad && sc2_conv_fused_gdn_supported(d), SC2_ERR_UNSUPPORTED,
                    "conv2d: fused GDN needs one tile to cover all output channels (Cout in {32,48,64,96}, or 256 on "
                    "the big-tile path: K >= 1024, >= 131072 output pixels, NHWC output); got Cout %d", d->Cout);
    SC2_REQUIRE(d->out_format >= SC2_OUT_BF16_NHWC && d->out_format <= SC2_OUT_I32_NCHW_SYM, SC2_ERR_INVALID_ARG,
                "conv2d: bad out_format");
    if (d->out_format == SC2_OUT_I32_NCHW_SYM)
        SC2_REQUIRE(d->epilogue == SC2_EPI_NONE && ep_beta && !scatter && sc2_conv_weight_rows(d->Cout) % 128 != 0,
                    SC2_ERR_INVALID_ARG, "conv2d: symbol output needs no epilogue, the medians in ep_beta and Cout <= 96");
    if (d->epilogue != SC2_EPI_NONE) SC2_REQUIRE(ep_beta, SC2_ERR_INVALID_ARG, "conv2d: epilogue needs ep_beta");
    if (epi_needs_x(d->epilogue) || fused)
        SC2_REQUIRE(ep_x, SC2_ERR_INVALID_ARG, "conv2d: epilogue needs ep_x");
    const long long M = (long long)d->N * OH * OW;
    SC2_REQUIRE(M < 0x7FFFFFFFLL, SC2_ERR_UNSUPPORTED, "conv2d: N*OH*OW = %lld exceeds 2^31", M);
    SC2_REQUIRE((long long)d->N * d->H < 0x7FFFFFFFLL, SC2_ERR_UNSUPPORTED, "conv2d: N*H too large");

    ConvArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_packed);
    a.y = y;
    a.ep_x = static_cast<const uint16_t *>(ep_x);
    a.ep_beta = ep_beta;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.SH = d->stride_h; a.SW = d->stride_w; a.PH = d->pad_h; a.PW = d->pad_w;
    a.OH = OH; a.OW = OW; a.OHW = OH * OW; a.M = (int)M;
    a.Kpad = d->Kpad; a.KT = 0; a.n_ntiles = 0;
    a.aop = d->a_op; a.epi = d->epilogue; a.out = d->out_format;
    a.g_pitch = sc2_conv_weight_pitch(d->Cout);
    {
        const unsigned long long xb = (unsigned long long)d->N * d->H * d->W * d->Cin * 2ull;
        const unsigned long long wb = (unsigned long long)d->Cout_pad * d->Kpad * 2ull;
        const bool fits = xb < 0x7FF00000ull && wb < 0x7FF00000ull;
        a.x_bytes = fits ? (unsigned)xb : 0u;
        a.w_bytes = fits ? (unsigned)wb : 0u;
    }
    a.k_slab_major = (d->k_order & SC2_K_SLAB_MAJOR) ? 1 : 0;
    a.b_kt_stride = (d->k_order & SC2_K_B_TILE_MAJOR) ? d->Cout_pad * 32 : 32;
    a.b_row_stride = (d->k_order & SC2_K_B_TILE_MAJOR) ? 32 : d->Kpad;
    a.dbg = sc2_pol().conv_debug;
    const bool needs_x_operand = epi_needs_x(d->epilogue);
    a.o_H = scatter ? d->out_H : 0; a.o_W = d->out_W; a.o_sh = d->out_stride_h; a.o_sw = d->out_stride_w;
    a.o_h0 = d->out_off_h; a.o_w0 = d->out_off_w;
    a.DH = dil_h; a.DW = dil_w;
    a.ep_x2 = static_cast<const uint16_t *>(ep_x2);
    a.y2 = y2;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (gdn_bwd) return d->Cout_pad == 96 ? launch<Gb_96>(a, s) : launch<Gb_128>(a, s);   // (geometry checked by sc2_gdn1_bwd_gemm)
    if (dilated) {
        // atrous convolution: the generic 128-wide tile with the dilation compiled in (every other instantiation -- static
        // geometries, window / patch staging, the 8-wave tiles -- derives input coordinates from undilated taps)
        SC2_REQUIRE(!fused && !scatter && d->Cout_pad % 128 == 0 && !(d->k_order & SC2_K_B_FRAG_MAJOR) &&
                        d->a_op != SC2_AOP_SQUARE && d->epilogue != SC2_EPI_GDN2 && d->epilogue != SC2_EPI_IGDN2 &&
                        d->out_format != SC2_OUT_I32_NCHW_SYM,
                    SC2_ERR_UNSUPPORTED, "conv2d: a dilated convolution needs Cout > 96 (packed rows %% 128 == 0), a dense output and "
                                         "one of the plain epilogues (got %d rows, epilogue %d)", d->Cout_pad, d->epilogue);
        return launch<Gd_128>(a, s);
    }

    const int rows = d->Cout_pad;
    if (d->k_order & SC2_K_B_FRAG_MAJOR) {
        // fragment-major weights are understood by the LDS-patch kernel only (sc2_conv_patch_supported tells)
        SC2_REQUIRE(sc2_conv_patch_supported(d), SC2_ERR_UNSUPPORTED,
                    "conv2d: SC2_K_B_FRAG_MAJOR weights need the 5x5 stride-2 patch geometry (Cin 96, Cout 48, OW <= 64, "
                    "bf16 NHWC output, slab-major K)");
        return launch_patch<C_conv2>(a, s);
    }
    // static geometries first (tile width must agree with the packed row count)
    if (rows == 96 && matches<C_conv0>(a)) return launch<C_conv0>(a, s);
    if (rows == 96 && matches<C_gdn96>(a)) return launch<C_gdn96>(a, s);
    if (rows == 48 && matches<C_conv2>(a)) return launch<C_conv2>(a, s);
    if (rows == 48 && matches<C_gdn48>(a)) return launch<C_gdn48>(a, s);
    if (rows == 32 && matches<C_conv4>(a)) return launch<C_conv4>(a, s);
    {
        // stride-1 layers with slab-major K: one staged window per 32-channel slab serves all taps (Cfg8::PATCH3).
        // Measured (MI355X, bs 256): 3x3 layers of the ResNet tail -24 % (layer2) / -22 % (layer3) with the 128-wide
        // tile; the 2x2 decoder layers +25 % SLOWER (their four taps re-read through L2 cheaply, the shifted reads cost
        // address arithmetic and LDS bank conflicts), so those stay on the im2col gather unless asked for.
        // SC2_CONV_PATCH3: 0 = off, 256 = the 256-wide tile for the 3x3 layers too, 2 = also the 2x2 decoder layers
        const int mode = sc2_pol().conv_patch3;
        const bool base_ok = mode != 0 && a.x_bytes != 0 && d->stride_h == 1 && d->stride_w == 1 && d->Cin % 32 == 0 &&
                             (d->k_order & SC2_K_SLAB_MAJOR) && !scatter && d->a_op == SC2_AOP_NONE;
        // rows of the window a 256-pixel tile can need: its own pixels, the taps' reach, and the drift of g(m) - m over
        // the output rows and image boundaries the tile crosses
        const long long drift_row = OW > d->W ? OW - d->W : d->W - OW;
        const long long hw = (long long)d->H * d->W, ohw = (long long)OH * OW;
        const long long drift_img = hw > ohw ? hw - ohw : ohw - hw;
        const long long window = 256 + (long long)(d->KH - 1) * d->W + d->KW + (256 / OW + 2) * drift_row +
                                 (256 / ohw + 1) * drift_img;
        if (base_ok && d->KH == 3 && d->KW == 3 && d->pad_h == 1 && d->pad_w == 1 && d->Cout % 128 == 0 &&
            d->out_format == SC2_OUT_BF16_NHWC && !fused && window <= P3_128::PATCH_ROWS) {
            if (d->Cout % 256 == 0 && mode == 256) return launch8<P3_256>(a, s);
            return launch8<P3_128>(a, s);
        }
        if (base_ok && mode == 2 && d->KH == 2 && d->KW == 2 && d->Cout == 256 && window <= P2_dec2::PATCH_ROWS &&
            big_tile_eligible(d, M, K)) {
            if (d->pad_h == 0 && d->pad_w == 0) return launch8<P2_dec2>(a, s);
            if (d->pad_h == 1 && d->pad_w == 1) return launch8<P2_dec4>(a, s);
        }
    }
    {
        // 3x3 stride-2 pad-1 layers: static-geometry 8-wave tile with buffer-addressed gather.  Measured (bs 256): the
        // 256-wide tile wins where the runtime-geometry big tile was already in use (layer3.0 conv2: 0.127 -> 0.107 ms); the
        // 128-wide one loses to the 4-wave kernel (layer2.0 0.143 -> 0.168 ms) and is only reachable with SC2_CONV_S2=128.
        const int mode = sc2_pol().conv_s2;
        if (mode != 0 && a.x_bytes != 0 && d->KH == 3 && d->KW == 3 && d->stride_h == 2 && d->stride_w == 2 && d->pad_h == 1 &&
            d->pad_w == 1 && d->Cin % 32 == 0 && !scatter && d->a_op == SC2_AOP_NONE && d->Cout % 128 == 0 && !fused) {
            if (mode == 128) return launch8<S2_128>(a, s);
            if (d->Cout % 256 == 0 && (M >= 256LL * 192 || mode == 256)) return launch8<S2_256>(a, s);
        }
    }
    if (d->a_op == SC2_AOP_SQUARE || d->epilogue == SC2_EPI_GDN2 || d->epilogue == SC2_EPI_IGDN2) {
        // squared-form GDN: its own two instantiations (Cfg::SQ)
        SC2_REQUIRE(rows % 128 == 0 && !scatter, SC2_ERR_UNSUPPORTED,
                    "conv2d: the squared-form GDN path needs Cout > 96 (packed rows %% 128 == 0), got %d rows", rows);
        if (needs_x_operand && d->out_format == SC2_OUT_BF16_NHWC) return launch<Gqx_128>(a, s);
        return launch<Gq_128>(a, s);
    }
    const bool big = big_tile_eligible(d, M, K);
    if (big && d->Cout % 256 == 0 && a.x_bytes == 0) return launch8<BG_256>(a, s);   // >= 2 GB: 64-bit addressing only
    if (big && d->Cout % 256 == 0) {
        if (sc2_pol().conv_half && d->out_format == SC2_OUT_BF16_NHWC) {
            if (matches<H_dec2>(a)) return launch8<H_dec2>(a, s);
            if (matches<H_dec4>(a)) return launch8<H_dec4>(a, s);
        }
        {
            if (sc2_pol().conv_big4 && d->out_format == SC2_OUT_BF16_NHWC) {   // A/B switch: the 4-wave register-tile kernel
                if (matches<R_dec2>(a)) return launch4<R_dec2>(a, s);
                if (matches<R_dec4>(a)) return launch4<R_dec4>(a, s);
                return launch4<RG_256>(a, s);
            }
        }
        {
            // persistent form of the two decoder layers (deferred output stores, conv_dec_persist.hip); SC2_CONV_PERSIST=0/1: A/B
            const int pmode = sc2_pol().conv_persist;   // 0: one workgroup per tile; 1: persistent; 2: + fragment reads a phase early; 3: one phase (32 MFMAs) per slab
            const bool persist = pmode && a.x_bytes != 0 && d->out_format == SC2_OUT_BF16_NHWC && !scatter &&
                                 d->Cout == 256 && d->a_op == SC2_AOP_NONE && (d->epilogue == SC2_EPI_NONE || fused);
#ifdef SC2_EXPERIMENTS   // timing experiment with garbage results: never in the shipped library
            if (persist && pmode == 32 && matches<B_dec2>(a)) return launch8p<B_dec2, 3>(a, s);
            if (persist && pmode == 32 && matches<B_dec4>(a)) return launch8p<B_dec4, 3>(a, s);
#else
            if (pmode == 32) return SC2_ERR_UNSUPPORTED;
#endif
            if (persist && pmode == 3 && matches<B_dec2>(a)) return launch8p<B_dec2, 2>(a, s);
            if (persist && pmode == 3 && matches<B_dec4>(a)) return launch8p<B_dec4, 2>(a, s);
            if (persist && pmode == 2 && matches<B_dec2>(a)) return launch8p<B_dec2, 1>(a, s);
            if (persist && pmode == 2 && matches<B_dec4>(a)) return launch8p<B_dec4, 1>(a, s);
            if (persist && matches<B_dec2>(a)) return launch8p<B_dec2, 0>(a, s);
            if (persist && matches<B_dec4>(a)) return launch8p<B_dec4, 0>(a, s);
        }
        if (matches<B_gdn512>(a)) return launch8<B_gdn512>(a, s);
        if (matches<B_dec2>(a)) return launch8<B_dec2>(a, s);
        if (matches<B_gdn256>(a)) return launch8<B_gdn256>(a, s);
        if (matches<B_dec4>(a)) return launch8<B_dec4>(a, s);
        return launch8<BG_256>(a, s);
    }
    if (big && d->Cout % 128 == 0) return launch8<BG_128>(a, s);
    bool epx = needs_x_operand &&
                     d->out_format == SC2_OUT_BF16_NHWC && !scatter && !sc2_pol().conv_no_epx;
    if (epx) {
        if (rows == 96 && matches<Cx_gdn96>(a)) return launch<Cx_gdn96>(a, s);
        if (rows == 48 && matches<Cx_gdn48>(a)) return launch<Cx_gdn48>(a, s);
        if (rows % 128 == 0) {
            if (matches<Cx_gdn512>(a)) return launch<Cx_gdn512>(a, s);
            if (matches<Cx_gdn256>(a)) return launch<Cx_gdn256>(a, s);
            return launch<Gx_128>(a, s);
        }
        if (rows == 96) return launch<Gx_96>(a, s);
        if (rows == 64) return launch<Gx_64>(a, s);
        if (rows == 48) return launch<Gx_48>(a, s);
        if (rows == 32) return launch<Gx_32>(a, s);
    }
    if (rows % 128 == 0) {
        if (matches<C_dec0>(a)) return launch<C_dec0>(a, s);
        if (matches<C_gdn512>(a)) return launch<C_gdn512>(a, s);
        if (matches<C_dec2>(a)) return launch<C_dec2>(a, s);
        if (matches<C_gdn256>(a)) return launch<C_gdn256>(a, s);
        if (matches<C_dec4>(a)) return launch<C_dec4>(a, s);
        return launch<G_128>(a, s);
    }
    if (rows == 96) return launch<G_96>(a, s);
    if (rows == 64) return launch<G_64>(a, s);
    if (rows == 48) return launch<G_48>(a, s);
    if (rows == 32) return launch<G_32>(a, s);
    sc2_set_error("conv2d: unsupported packed row count %d", rows);
    return SC2_ERR_UNSUPPORTED;
}
}  // namespace
