// vp_runtime.h -- process-wide libvphip context used by the vplib templates.
// The reference pins device 0 and the default stream in main() (apps/cli/main.cpp:22-23); here the
// context is created on first GPU use, so CPU-only variants (-t 0, -t 3) never touch the device.
#ifndef VPLIB_RUNTIME_H
#define VPLIB_RUNTIME_H

#include "vphip.h"

namespace vplib {

// Device used by Context(); call before the first GPU operation (default 0).
void SetDevice(int device);

// Creates the context on first call; prints the reference-style assert line and exits on failure.
vp_ctx* Context();

// Destroys the context (optional; also done at process exit).
void Shutdown();

}  // namespace vplib

#endif
