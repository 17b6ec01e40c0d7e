// Kernel-side argument block shared by the two convolution kernels (conv_igemm.hip, conv_wino.hip).
#pragma once
#include "common.h"

namespace idiff_detail {

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
    int tiles_x, ntiles, ncob;
    unsigned total_wg;
};

// true when the Winograd F(2x2,3x3) kernel covers this problem (3x3, whole 8x32 patches, Cin % 8 == 0, Cout % 64 == 0)
bool conv_wino_eligible(const ConvArgs& a, int ks, int mode);
// launches it; a.tiles_x / ntiles / ncob / total_wg must describe 8x32-pixel patches and 64-channel blocks
int launch_conv_wino(const ConvArgs& a, int mode, hipStream_t st);

// F(4x4,3x3) kernel (conv_wino4.hip): 3x3, H and W multiples of 4, Cin % 8 == 0, Cout % 16 == 0, at least one 16x32-pixel x
// 64-channel item per CU; a.tiles_x / ntiles / ncob describe 8x32-pixel patches (the GroupNorm-partials grid) as above
// requested: the caller asked for this kernel by name (idiff_conv_desc.algo_request): the items-per-sample threshold is waived
bool conv_wino4_eligible(const ConvArgs& a, int ks, int mode, bool requested = false);
int launch_conv_wino4(const ConvArgs& a, int mode, hipStream_t st);

}  // namespace idiff_detail
