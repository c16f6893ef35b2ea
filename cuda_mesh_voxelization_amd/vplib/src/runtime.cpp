// runtime.cpp -- lazily created libvphip context shared by the vplib templates.
#include "vp_runtime.h"

#include <cstdio>
#include <cstdlib>

#include "debug_utils.h"

namespace vplib {

namespace {
int g_device = 0;
vp_ctx* g_ctx = nullptr;
}  // namespace

void SetDevice(int device) { g_device = device; }

vp_ctx* Context()
{
    if (!g_ctx) {
        gpuAssert(vp_ctx_create(g_device, &g_ctx));
        std::atexit(Shutdown);
    }
    return g_ctx;
}

void PrintDeviceTimes(const std::string& label)
{
    if (!g_ctx) return;
    for (int k = 0; k < VP_K_COUNT; ++k) {
        double ms = 0; uint64_t launches = 0;
        if (vp_prof_get(g_ctx, k, &ms, &launches) == 0 && launches)
            std::printf("# device-time %s %s %f ms %llu launches\n", label.c_str(), vp_prof_name(k), ms, (unsigned long long)launches);
    }
}

void Shutdown()
{
    if (g_ctx) {
        vp_ctx_destroy(g_ctx);
        g_ctx = nullptr;
    }
}

}  // namespace vplib
