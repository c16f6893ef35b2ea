// vp_runtime.h -- process-wide libvphip context used by the vplib templates.
// The reference pins device 0 and the default stream in main() (apps/cli/main.cpp:22-23); here the
// context is created on first GPU use, so CPU-only variants (-t 0, -t 3) never touch the device.
#ifndef VPLIB_RUNTIME_H
#define VPLIB_RUNTIME_H

#include <string>
#include <vector>

#include "vphip.h"

namespace vplib {

// Device used by Context(); call before the first GPU operation (default 0).
void SetDevice(int device);

// More than one device: every GPU Compute() (VOX / CSG / JFA, Types::NAIVE and Types::TILED) cuts the grid into Z-slabs, one
// per entry of `devices`, through vp_multi_* (include/vphip.h) -- what stands where the reference pins device 0
// (apps/cli/main.cpp:22-23).  `ghost` picks the JFA variant: true = recomputed ghost planes, no exchange between passes
// (VP_MULTI_GHOST); false = halo planes copied device to device before every pass (VP_MULTI_HALO).  Results are bit-identical
// to the single-device path.  A device may be named several times (several contexts on it).  Call before the first GPU operation.
void SetDevices(const std::vector<int>& devices, bool ghost = true);

// Devices visible to the process (0 when there is none or the runtime fails).
int DeviceCount();

// The slab driver, created on first call; nullptr while fewer than two devices are set.
vp_multi* Multi();
int MultiMode();
// VP_MULTI_HALO / VP_MULTI_GHOST / VP_MULTI_HYBRID / VP_MULTI_TRANSPOSE (include/vphip.h) for the next JFA on several devices; call after SetDevices.
void SetMultiMode(int mode);

// Creates the context on first call; prints the reference-style assert line and exits on failure.
vp_ctx* Context();

// Destroys the context (optional; also done at process exit).
void Shutdown();

// Workspace slots of the shared context (vp_ctx_workspace) the Compute() wrappers keep their device buffers in.
enum { kSlotGridA = 0, kSlotGridB = 1, kSlotXyz = 2, kSlotTri = 3, kSlotSdf = 4, kSlotRecords = 5 };

// PROFILING builds: prints the device time (hipEvents, vp_prof_*) of every kernel that ran since the last vp_prof_reset as
// "# device-time <label> <kernel> <ms> ms <n> launches" lines.  Deliberately NOT the "[label]: x ms" grammar of
// PROFILING_SCOPE (profiling.h): parsers of the reference's benchmark contract must not see extra columns.
void PrintDeviceTimes(const std::string& label);

// The same for the slab driver (-g G, SetDevices): MultiProfile(true) resets and enables the timers of every rank's context,
// MultiProfile(false) disables them; PrintMultiDeviceTimes prints, per kernel, the time of the SLOWEST rank (ranks run side by
// side, so that is what the job waits for) as "# device-time <label> <kernel> <ms> ms <n> launches max-over-<G>-devices";
// MultiDeviceTime returns that maximum for one kernel key.
void MultiProfile(bool on);
void PrintMultiDeviceTimes(const std::string& label);
double MultiDeviceTime(int kernel);

}  // namespace vplib

#endif
