// The one place libkpx_hip.so reads the environment (see kpx_env.h).
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "kpx_env.h"
#include "../../include/kpx.h"

static KpxEnv g_env;
static std::atomic<int> g_env_loaded{0};
static std::mutex g_env_mutex;

static int env_flag(const char* name) { return getenv(name) != nullptr; }
static long env_long(const char* name, long dflt) { const char* v = getenv(name); return v ? atol(v) : dflt; }

static void env_parse(KpxEnv* e) {
    e->no_wino = env_flag("KPX_NO_WINO");
    e->no_wino43 = env_flag("KPX_NO_WINO43");
    e->no_wino_wgrad = env_flag("KPX_NO_WINO_WGRAD");
    e->no_c16 = env_flag("KPX_NO_C16");
    e->no_rgb = env_flag("KPX_NO_RGB");
    e->no_splitk = env_flag("KPX_NO_SPLITK");
    e->no_smallcout = env_flag("KPX_NO_SMALLCOUT");
    e->no_wrows = env_flag("KPX_NO_WROWS");
    e->no_wmerge = env_flag("KPX_NO_WMERGE");
    e->no_wtaprows = env_flag("KPX_NO_WTAPROWS");
    e->wgrad_4w = env_flag("KPX_WGRAD_4W");
    e->no_merge_kh = (int)env_long("KPX_NO_MERGE", 0);
    e->tile_bm = e->tile_bn = 0;
    if (const char* ov = getenv("KPX_TILE")) { int a = 0, b = 0; if (sscanf(ov, "%d,%d", &a, &b) == 2) { e->tile_bm = a; e->tile_bn = b; } }
    e->splitk_maxtiles = env_long("KPX_SPLITK_MAXTILES", 256);
    e->wgrad_target = env_long("KPX_WGRAD_TARGET", 0);
    e->wino_kmin = (int)env_long("KPX_WINO_KMIN", 4);
    e->wino_nmin = (int)env_long("KPX_WINO_NMIN", 4);
    e->wino_ct = (int)env_long("KPX_WINO_CT", 0);
    e->wino_stagger = (int)env_long("KPX_WINO_STAGGER", 1);
    e->ww_comin = (int)env_long("KPX_WW_COMIN", 4);
    e->ww_target = env_long("KPX_WW_TARGET", 256);
    e->bf16s_variant = (int)env_long("KPX_BF16S_VARIANT", 0);
    e->gauss_blocks = (int)env_long("KPX_GAUSS_BLOCKS", 0);          // 0: sized from the tensor (keypoints.hip)
    e->gauss_nt = (int)env_long("KPX_GAUSS_NT", 1);
    e->no_gemm3 = env_flag("KPX_NO_GEMM3");
    e->no_wgrad3 = env_flag("KPX_NO_WGRAD3");
    e->wgrad3_first = (int)env_long("KPX_WGRAD3_FIRST", 0);
    e->no_wsmall = env_flag("KPX_NO_WSMALL");
    e->no_wsmall32 = env_flag("KPX_NO_WSMALL32");
    e->wsmall_c64_max = (int)env_long("KPX_WSMALL_C64_MAX", 4);
}

extern "C" __attribute__((visibility("hidden"))) const KpxEnv* kpx_env() {
    if (!g_env_loaded.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_env_mutex);
        if (!g_env_loaded.load(std::memory_order_relaxed)) {
            env_parse(&g_env);
            g_env_loaded.store(1, std::memory_order_release);
        }
    }
    return &g_env;
}

// Re-read the environment.  NOT thread-safe against concurrent launches: call it between launches from the launching thread.
extern "C" int kpx_reload_env(void) {
    std::lock_guard<std::mutex> lock(g_env_mutex);
    env_parse(&g_env);
    g_env_loaded.store(1, std::memory_order_release);
    return 0;
}
