// Switches of libkpx_hip.so.  The environment is parsed ONCE, when the library is first used (include/kpx.h: "read once"); kpx_reload_env()
// (exported) parses it again -- for harnesses that flip a switch between launches (bench.py times the direct kernel with KPX_NO_WINO=1).  No
// launch path calls getenv().  Three switches, each flipped by a test or a bench leg (DESIGN.md lists every KPX_* variable of the repo):
#pragma once

struct KpxEnv {
    int no_wino;            // KPX_NO_WINO: the 3x3 stride-1 layers on the direct implicit-GEMM kernels (bench roofline_direct_conv, tests)
    int no_gemm3;           // KPX_NO_GEMM3: the fp32-MFMA kernels instead of the bf16x3 ones (bench roofline_direct_conv, float64 tests)
    int bf16s_variant;      // KPX_BF16S_VARIANT: v + 1 forces tile variant v of conv3x3_bf16s_kernel (scratch/bf16s_check.py sweeps)
};

extern "C" __attribute__((visibility("hidden"))) const KpxEnv* kpx_env();
