// runtime.cpp -- lazily created libvphip context shared by the vplib templates.
#include "vp_runtime.h"

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

void Shutdown()
{
    if (g_ctx) {
        vp_ctx_destroy(g_ctx);
        g_ctx = nullptr;
    }
}

}  // namespace vplib
