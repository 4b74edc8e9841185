// api_voxel.hip -- the device voxel grid (ccal_downsample_raw, ccal_download_cloud).
#include "ccal_internal.hpp"
#include "kernels_common.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <chrono>
#include <limits>
#include <new>
#include <thread>

using namespace ccal;

#include "voxel_impl.hpp"
