// Utils/SpatialInertia.h -- same include path as the reference (include/grbda/Utils/SpatialInertia.h); the facade lives in grbda.h
#pragma once
#include "../grbda.h"
