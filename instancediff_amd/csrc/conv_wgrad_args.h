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
};

bool wino_wgrad_eligible(const WwArgs& a, int ks, int mode);
void wino_wgrad_geometry(int Cin, int Cout, int B, int Hout, int Wout, int* ncob, int* ncib, int* nsplit);
int launch_wino_wgrad(const WwArgs& a, int mode, hipStream_t st);

}  // namespace idiff_detail
