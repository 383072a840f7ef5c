// libfidget_hip.so: C ABI (include/fidget_hip.h) + host driver of the render pipeline.
// Single translation unit: the kernels are included so that one `hipcc -shared` builds
// everything for gfx950.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <functional>
#include <vector>

#include "../../include/fidget_hip.h"
#include "../../include/fidget_hip_debug.h"
#include "host_graph.hpp"
#include "host_regtape.hpp"
#include "kernels.hip"
#include "effects.hip"
#include "mesh.hip"
#include "prune2.hip"
#include "host_mesh.hpp"

#define FH_LDS_MAX 163840  // 160 KiB per workgroup on gfx950

// The C ABI as one translation unit in eight fragments (each was a section of this file when it was 2 900 lines long): the fragments are
// not stand-alone headers - they are included here, in this order, and share the static helpers of capi_core.hpp.
#include "capi_core.hpp"

extern "C" {

#include "capi_context.hpp"
#include "capi_tapes.hpp"
#include "capi_eval.hpp"
#include "capi_render.hpp"
#include "capi_effects.hpp"
#include "capi_mesh.hpp"
#include "capi_debug.hpp"

}  // extern "C"
