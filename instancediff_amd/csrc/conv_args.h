// Kernel-side argument block shared by the two convolution kernels (conv_igemm.hip, conv_wino.hip).
#pragma once
#include "common.h"

namespace idiff_detail {

// arguments of the GroupNorm finalize that rides on a conv call (idiff_conv_desc.gn_*; the finalize is a launch behind the conv)
struct GnTail {
    const float* gamma;
    const float* beta;
    const float* film;
    long long film_ld;
    float eps;
    int groups;
    float* out_a;
    float* out_b;
    float* mean_rstd;
};

struct ConvArgs {
    const float* src0;
    const float* src1;
    long long bs0, bs1;
    int C0v, C1v;  // virtual channel counts (x4 for unshuffle)
    int C0r;       // real channel count of src0 (prologue tables are indexed by real channel)
    int Cin;       // C0v + C1v
    int B, Hin, Win, Hout, Wout;
    int Cout;
    const float* wpk;
    const float* wwino;  // Winograd-domain weights (idiff_pack_conv_weight_wino) or null
    const float* wwino4; // F(4x4,3x3)-domain weights (idiff_pack_conv_weight_wino4) or null
    const float* bias;
    const float* pro_a;
    const float* pro_b;
    float* out;
    long long obs;
    const float* res;
    long long rbs;
    const float* vec;
    const float* aux;
    long long abs_;
    const float* aux_a;
    const float* aux_b;
    float* stats;
    GnTail gn;
    int tiles_x, ntiles, ncob;
    unsigned total_wg;
};

// true when the Winograd F(2x2,3x3) kernel covers this problem (3x3, whole 8x32 patches, Cin % 8 == 0, Cout % 64 == 0)
bool conv_wino_eligible(const ConvArgs& a, int ks, int mode);
// launches it; a.tiles_x / ntiles / ncob / total_wg must describe 8x32-pixel patches and 64-channel blocks
int launch_conv_wino(const ConvArgs& a, int mode, hipStream_t st);

// F(4x4,3x3) kernels: H and W multiples of 4, W >= 24, Cin % 8 == 0, Cout % 16 == 0, a weight image (idiff_pack_conv_weight_wino4).
// conv_wino4_items: 0 when they do not cover the problem, else the 16x32-pixel x 64-channel items PER SAMPLE (the quantity the choice
// between the kernels is made on -- never the batch).  `requested`: asked for by name (idiff_conv_desc.algo_request), which also
// overrides IDIFF_WINOGRAD4=0.  a.tiles_x / ntiles / ncob describe 8x32-pixel patches (the GroupNorm-partials grid) as above.
long long conv_wino4_items(const ConvArgs& a, int ks, int mode, bool requested = false);
// conv_wino4.hip: one 512-thread workgroup per CU, items of 16x32 pixels x 64 channels, weights staged through LDS
int launch_conv_wino4(const ConvArgs& a, int mode, hipStream_t st);
// conv_wino4h.hip: two 256-thread workgroups per CU, items of 8x32 pixels x 64 channels, weights read straight into the A operand
int launch_conv_wino4h(const ConvArgs& a, int mode, hipStream_t st);

// conv1x1_x3.hip: 1x1 conv on the bf16 matrix cores, fp32 operands split three ways (six MFMAs per product: fp32-class result).
// `a` must describe the flattened 1x1 geometry (Hout == 1, 256-pixel tiles, 64-channel blocks); eligible: Wout % 256 == 0,
// Cout % 64 == 0, C0v % 8 == 0, Cin % 8 == 0, Cin >= 32, no prologue, no GroupNorm partials.  wx3 = image of idiff_pack_conv1x1_x3.
// Unshuffle mode (pixel-unshuffle downsample): the real geometry, single source, Wout % 4 == 0, output pixels per sample % 256 == 0.
bool conv1x1_x3_eligible(const ConvArgs& a, int mode);
int launch_conv1x1_x3(const ConvArgs& a, int mode, const void* wx3, hipStream_t st);

}  // namespace idiff_detail
