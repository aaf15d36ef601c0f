// fspann_api.hip — implementation of include/fspann.h for gfx950 (MI355X).
// Product code.  No CPU fallback exists: every compute entry point launches a
// HIP kernel or fails with FSPANN_E_DEVICE.  Nothing here references oracle/.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>
#include <type_traits>
#include <mutex>
#include <new>
#include <numeric>

#include "comm.hip.h"
#include "build.hip.h"
#include "encode.hip.h"
#include "groundtruth.hip.h"
#include "hostpipe.hip.h"
#include "refine.hip.h"
#include "route.hip.h"
#include "route_lazy.hip.h"
#include "tick.hip.h"
#include "../host/route_replay.hpp"


// The entry points, by stage (one translation unit: the kernels above are templates shared by several of them).
#include "api_common.hip.h"
#include "api_encode.hip.h"
#include "api_setup.hip.h"
#include "api_route.hip.h"
#include "api_refine.hip.h"
#include "api_tick.hip.h"
#include "api_hostpipe.hip.h"
#include "api_misc.hip.h"
#include "api_build.hip.h"
#include "api_ext.hip.h"
