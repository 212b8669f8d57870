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
    e->no_gemm3 = env_flag("KPX_NO_GEMM3");
    e->bf16s_variant = (int)env_long("KPX_BF16S_VARIANT", 0);
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
