// Geometry of the "gather convolution" shared by the fp32 implicit-GEMM kernels (conv_igemm.hip) and the bf16x3 kernels (conv_gemm3.hip):
//   out[n, a*osy+oy0, b*osx+ox0, j] = act( bias[j] + sum_{tr,tq,c} in[n, a*isy+tr*ity+iy0, b*isx+tq*itx+ix0, c] * B[tap(tr,tq)][c][j] )
// forward: in = x, B = HWIO weights; dgrad: in = dy, B = the same weights read transposed, one class per stride parity of dx pixels.
#pragma once
#include "kpx_common.h"

// per stride-parity class of output pixels (dgrad of a strided conv); forward has exactly one class
struct ConvClass { int Ha, Wa, oy0, ox0, Tr, Tq, iy0, ix0, wr0, wq0, M, mt; };

struct ConvGeom {
    const float* x; float* y; const float* w; const float* bias;
    int N, Hi, Wi, Cin, ldx;
    int Ho, Wo, Cout, ldy;
    int Ha, Wa;
    int osy, oy0, osx, ox0;
    int isy, iy0, isx, ix0;
    int Tr, Tq, ity, itx;
    int wr0, wrs, wq0, wqs, KW;
    int wts, ldw;
    int act, vecA, vecB;
    int M, mt, nt;
    ConvClass cls[4]; int ncls;   // blockIdx.y selects the class
    int ksplit; float* ws;        // split-K: blockIdx.z owns an equal slice of the K chunks; raw partials go to ws[z][pixel][Cout]
    size_t ws_slab;               // floats per split slab = N*Ho*Wo*Cout
    const float* mul_y; int ld_mul, mul_act;   // optional epilogue factor act'(mul_y[pixel][col]) (data gradient of an activated tensor: kpx_conv2d_dgrad_act_f32)
    int terms;      // bf16 terms per fp32 operand on the bf16-pipe kernels (conv_gemm3.hip): 3 = fp32-equivalent, 1 = bf16 operands (the `arith` argument)
    int merge;      // >0: row-merged taps for tiny Cin (= original Cin): the KW*Cin floats of one filter row are
                    // contiguous in NHWC, so they are treated as one tap with KW*Cin channels (per-element x bounds)
    int io16;       // bf16 configuration (conv_gemm3.hip only, terms == 1): bit 0: x is bf16, bit 1: y is bf16, bit 2: mul_y is bf16
                    // (the pointers above are then bf16 tensors behind their float* type; pixel strides count ELEMENTS either way)
};


// Weight gradient dw[tap][c][k] = sum_p x[p shifted by tap][c] * dy[p][k]: GEMM with M = Cin, N = Cout, K = pixels, split over workgroups
// along K (fixed-order reduction of the partial slabs afterwards: bitwise reproducible, no float atomics).
struct WgradGeom {
    const float* x; const float* dy; float* out;
    int N, Hi, Wi, Cin, ldx;
    int Ho, Wo, Cout, lddy;
    int KH, KW, stride, pad_t, pad_l;
    int P, S, pps;          // pixels, splits, pixels per split (multiple of 16)
    int ct, kt;             // channel tiles
    int vecA, vecB;
    int terms;              // see ConvGeom::terms
    int merge;              // >0: row-merged taps (see ConvGeom::merge); Cin/KW below are the merged values
    size_t slab;            // floats per slab = KH*KW*Cin*Cout
    int io16;               // bf16 configuration (conv_gemm3.hip only, terms == 1): x AND dy are bf16 tensors
};
