// runtime.cpp -- lazily created libvphip context shared by the vplib templates.
#include "vp_runtime.h"

#include <cstdio>
#include <cstdlib>

#include "debug_utils.h"

namespace vplib {

namespace {
int g_device = 0;
vp_ctx* g_ctx = nullptr;
std::vector<int> g_devices;
bool g_ghost = true;
int g_mode = -1;                                                   // SetMultiMode: overrides the ghost / halo choice of SetDevices
vp_multi* g_multi = nullptr;
}  // namespace

void SetDevice(int device) { g_device = device; }

void SetDevices(const std::vector<int>& devices, bool ghost)
{
    if (g_multi && devices != g_devices) {                         // another device list: the slab driver is rebuilt on next use
        vp_multi_destroy(g_multi);
        g_multi = nullptr;
    }
    g_devices = devices;
    g_ghost = ghost;
    g_mode = -1;
    if (!devices.empty()) g_device = devices[0];                   // exports and single-device calls use the first one
}

vp_multi* Multi()
{
    if (g_devices.size() < 2) return nullptr;
    if (!g_multi) {
        gpuAssert(vp_multi_create(g_devices.data(), static_cast<int>(g_devices.size()), &g_multi));
        static bool registered = false;
        if (!registered) { std::atexit(Shutdown); registered = true; }
    }
    return g_multi;
}

int MultiMode() { return g_mode >= 0 ? g_mode : (g_ghost ? VP_MULTI_GHOST : VP_MULTI_HALO); }

void SetMultiMode(int mode) { g_mode = mode; }

int DeviceCount()
{
    int n = 0;
    return vp_device_count(&n) == 0 ? n : 0;
}

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

void MultiProfile(bool on)
{
    if (!g_multi) return;
    for (int r = 0; r < vp_multi_count(g_multi); ++r) {
        vp_ctx* c = vp_multi_ctx(g_multi, r);
        if (on) gpuAssert(vp_prof_reset(c));
        gpuAssert(vp_prof_enable(c, on ? 1 : 0));
    }
}

namespace {
// slowest rank of kernel k: its time and its launch count
bool multi_time(int k, double& ms, uint64_t& launches)
{
    ms = 0; launches = 0;
    if (!g_multi) return false;
    for (int r = 0; r < vp_multi_count(g_multi); ++r) {
        double t = 0; uint64_t c = 0;
        if (vp_prof_get(vp_multi_ctx(g_multi, r), k, &t, &c) == 0 && c && t >= ms) { ms = t; launches = c; }
    }
    return launches != 0;
}
}  // namespace

double MultiDeviceTime(int kernel)
{
    double ms = 0; uint64_t launches = 0;
    multi_time(kernel, ms, launches);
    return ms;
}

void PrintMultiDeviceTimes(const std::string& label)
{
    if (!g_multi) return;
    for (int k = 0; k < VP_K_COUNT; ++k) {
        double ms = 0; uint64_t launches = 0;
        if (multi_time(k, ms, launches))
            std::printf("# device-time %s %s %f ms %llu launches max-over-%d-devices\n", label.c_str(), vp_prof_name(k), ms,
                        (unsigned long long)launches, vp_multi_count(g_multi));
    }
}

void Shutdown()
{
    if (g_multi) {
        vp_multi_destroy(g_multi);
        g_multi = nullptr;
    }
    if (g_ctx) {
        vp_ctx_destroy(g_ctx);
        g_ctx = nullptr;
    }
}

}  // namespace vplib
