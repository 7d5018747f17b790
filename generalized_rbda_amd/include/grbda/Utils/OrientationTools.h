// Utils/OrientationTools.h -- same include path as the reference (include/grbda/Utils/OrientationTools.h); the facade lives in grbda.h
#pragma once
#include "../grbda.h"
