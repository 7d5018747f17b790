// Dynamics/Body.h -- same include path as the reference (include/grbda/Dynamics/Body.h); the facade lives in grbda.h
#pragma once
#include "../grbda.h"
