// Argument block of the Winograd weight-gradient kernel (conv_wino_wgrad.hip), filled by idiff_conv2d_wgrad (conv_wgrad.hip).
#pragma once
#include "common.h"

namespace idiff_detail {

struct WwArgs {
    const float* src0;
    const float* src1;
    long long bs0, bs1;
    int C0v, C1v, C0r, Cin;
    int B, Hin, Win, Hout, Wout;
    int Cout;
    const float* pro_a;
    const float* pro_b;
    const float* dy;
    long long dybs;
    float* ws;  // [nsplit][9][Cin][Cout]
    int ncob, ncib, nsplit;
    int ups;  // F(4x4,3x3) form only: the input is the nearest x2 upsample of src0 (IDIFF_CONV_UPSAMPLE2)
};

bool wino_wgrad_eligible(const WwArgs& a, int ks, int mode);
void wino_wgrad_geometry(int Cin, int Cout, int B, int Hout, int Wout, int* ncob, int* ncib, int* nsplit);
int launch_wino_wgrad(const WwArgs& a, int mode, hipStream_t st);

// conv_wino4_wgrad.hip: the F(4x4,3x3) form (normal and upsample mode; Hout % 4 == 0, Wout % 16 == 0, Cout % 64 == 0, Cin % 16 == 0); blocks of
// 64 co x 32 ci, same ws layout [nsplit][9][Cin][Cout]
bool wino4_wgrad_eligible(const WwArgs& a, int ks, int mode);
void wino4_wgrad_geometry(int Cin, int Cout, int B, int Hout, int Wout, int* ncob, int* ncib, int* nsplit);
int launch_wino4_wgrad(const WwArgs& a, hipStream_t st);

}  // namespace idiff_detail
