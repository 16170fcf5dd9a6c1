// fused_device.hpp — what the fused kernels (fused_engine.hip, sumchain_engine.hip) share: the device helpers plus their plan structs.
#pragma once
#include "device_util.hpp"
#include "fused_plan.hpp"
