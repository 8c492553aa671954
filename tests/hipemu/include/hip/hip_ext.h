// hipemu: hipExtLaunchKernelGGL lives in hip_runtime.h
#pragma once
#include "hip_runtime.h"
