// Tuning / debugging switches of libkpx_hip.so.  The environment is parsed ONCE, when the library is first used (include/kpx.h: "read
// once"); kpx_reload_env() (exported) parses it again -- for harnesses that flip a switch between launches (bench.py times the direct kernel
// with KPX_NO_WINO=1).  No launch path calls getenv().
#pragma once

struct KpxEnv {
    int no_wino, no_wino43, no_wino_wgrad, no_c16, no_rgb, no_splitk, no_smallcout, no_wrows, no_wmerge, no_wtaprows, wgrad_4w;
    int no_merge_kh;            // KPX_NO_MERGE=<KH>: no row merging for filters of that height (0: merge everywhere)
    int tile_bm, tile_bn;       // KPX_TILE="BM,BN" (0,0: planner's choice)
    long splitk_maxtiles;       // KPX_SPLITK_MAXTILES (256)
    long wgrad_target;          // KPX_WGRAD_TARGET (0: per-tile default)
    int wino_kmin, wino_nmin, wino_ct, wino_stagger;
    int ww_comin; long ww_target;
    int bf16s_variant;
    int gauss_blocks, gauss_nt;
    int no_gemm3, no_wgrad3, wgrad3_first;  // bf16x3 implicit-GEMM family (conv_gemm3.hip)
    int no_wsmall, no_wsmall32, wsmall_c64_max;                       // tiny-filter weight gradients (conv_wsmall.hip); Cout limit of its 64-channel 3x3 variant
};

extern "C" __attribute__((visibility("hidden"))) const KpxEnv* kpx_env();
