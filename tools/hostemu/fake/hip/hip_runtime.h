// Shadows <hip/hip_runtime.h> in the host-emulation build (tools/hostemu/build.sh); everything comes from hip_shim.h.
#pragma once
#include "../../hip_shim.h"
