// prl_all.hpp -- what every translation unit of the library includes first: the HIP runtime, the C ABI, the launcher
// declarations and the device headers every kernel needs (prl_device: descriptor + wave helpers + reference
// arithmetic, prl_ray, prl_search, prl_paint, prl_observe, prl_state, prl_step).  All device code is float64 in the
// reference's operation order (oracle/paint_oracle.c is the scalar statement; numpy.dot -> explicit fma chain,
// everything else unfused): compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "paintrl.h"
#include "prl_launch.hpp"
#include "prl_dynlds.hpp"

#include "prl_diag.hpp"
#include "prl_device.hpp"
#include "prl_ray.hpp"
#include "prl_search.hpp"
#include "prl_paint.hpp"
#include "prl_observe.hpp"
#include "prl_state.hpp"
#include "prl_step.hpp"
