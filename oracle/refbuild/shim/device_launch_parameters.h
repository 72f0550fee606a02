// TEST INFRASTRUCTURE ONLY. Stand-in for device_launch_parameters.h; see cuda_runtime.h.
#pragma once
#include "cuda_runtime.h"
