// Utils/SpatialTransforms.h -- same include path as the reference (include/grbda/Utils/SpatialTransforms.h); the facade lives in grbda.h
#pragma once
#include "../grbda.h"
